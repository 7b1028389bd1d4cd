"""GPU idle time between kernels of the training steps, from a rocprofv3 --kernel-trace CSV:
  python tools/trace_gaps.py <kernel_trace.csv> [first_step last_step]
A step = one launch of the fused backward (project_bwd1_kernel<true, false>); the window runs from
step `first_step` to step `last_step` (default 10..30), overlapping kernels are merged.
Round 1, S2: 15 gaps in 900 kernels, the largest inside a step 6 us: the step is kernel-bound and a
hipGraph would have nothing to remove."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "project_bwd1_kernel<true, false>" in r["Kernel_Name"]]
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10, 30)
if len(marks) > b:
    rows = rows[marks[a] + 1: marks[b] + 1]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, cur_s, cur_e = 0, None, None
gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, r["Kernel_Name"][:60]))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
wall = t1 - t0
print(f"kernels {len(rows)}  wall {wall/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {100*(wall-busy)/wall:.1f} %  mean gap {sum(g for g,_ in gaps)/max(len(gaps),1)/1e3:.2f} us over {len(gaps)} gaps")
gaps.sort(reverse=True)
for g, name in gaps[:8]:
    print(f"  {g/1e3:8.1f} us before {name}")
