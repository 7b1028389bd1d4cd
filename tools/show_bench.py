import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(d[k], 3) for k in ('value', 'ms_per_step', 'render_fps')}, d['config']['intersections'], d['config']['visible'])
print('roofline', d['roofline'])
if d.get('cpu_baseline'): print('cpu', d['cpu_baseline']['value'])
for k, v in list(d['stages'].items())[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{k:22s} {v['ms_per_step']:.3f} ms/step  {v['launches_per_step']:4.1f} x {v['us_per_launch']:8.1f} us  {v.get('alg_GBps', 0):8.1f} GB/s")
