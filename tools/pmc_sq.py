"""Per-kernel means of arbitrary rocprofv3 --pmc counters: python tools/pmc_sq.py a.csv [b.csv ...] [--k name,...]"""
import csv, re, sys
from collections import defaultdict
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("mfma_raster::", "").replace("void ", "")
    base = re.sub(r"_kernel$", "", re.sub(r"[<(].*", "", n))
    if base in ("rasterize_bwd_mm", "project_bwd1"):          # the instantiations are different kernels: <HAS_BG, ABSGRAD, EXP, Shape> / <FUSE>
        m = re.search(r"<([^>]*)>", n)
        if m:
            base += "<" + m.group(1).replace("mfma_raster::", "").replace(" ", "") + ">"
    return base
files = [a for a in sys.argv[1:] if not a.startswith("--")]
want = None
for a in sys.argv[1:]:
    if a.startswith("--k="): want = a[4:].split(",")
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if int(r["Grid_Size"]) < 1_000_000 and k.startswith("rs_"): k += "/small"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    if want and k not in want: continue
    if not want and (k.startswith("at::") or k.startswith("__amd")): continue
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
