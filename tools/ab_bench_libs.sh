#!/bin/bash
# bench.py through several builds of the library on one box, alternating: tools/ab_bench_libs.sh <outdir> <lib> [<lib> ...]
# ("product" = the in-tree library).  One preset, no CPU baseline; prints it/s and the stages that matter.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
for rep in 1 2; do for lib in "$@"; do
  name=$(basename $lib .so)_$rep
  if [ "$lib" = product ]; then unset MI3DGS_LIB; else export MI3DGS_LIB=$R/$lib; fi
  timeout -k 10 200 python3 $R/bench.py --one-preset --no-cpu-baseline --steps 40 $BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err || { tail -3 $OUT/$name.err; exit 1; }
  python3 - $OUT/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages"]
g=lambda k: round(s[k]["us_per_launch"],1) if k in s else None
print("%-24s %7.1f it/s (no refine %7.1f)  fwd %s  loss %s+%s  culled-adam %s  bwd %s  pbwd_adam %s  pfwd %s emit %s" % (sys.argv[2], d["value"], d["presets"][d["config"]["preset"]]["it_per_s_without_refine"], g("rasterize_fwd"), g("loss_fwd"), g("loss_bwd"), g("adam_culled_groups"), g("rasterize_bwd"), g("project_bwd_adam"), g("project_fwd"), g("tile_emit")))
PY
done; done
