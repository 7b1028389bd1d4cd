#!/bin/bash
# same-box A/B of the round-4 rasterize_bwd variants (tools/build_bwd_variants.sh): S2 plain / absgrad, wolf 960x720 with segments
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r04_ab}; mkdir -p $OUT; shift
LIBS=""; for v in "$@"; do LIBS="$LIBS $R/tools/ab/libmi3dgs_v$v.so"; done
N=$#
SEG=$(python3 -c "print(' '.join(['1']*$N))")
timeout -k 10 200 python3 $R/tools/raster_ab.py --libs $LIBS --scene garden --reps 12 > $OUT/garden_plain.json 2> $OUT/garden_plain.err || { tail -5 $OUT/garden_plain.err; exit 1; }
timeout -k 10 200 python3 $R/tools/raster_ab.py --libs $LIBS --scene garden --absgrad --reps 12 > $OUT/garden_abs.json 2> $OUT/garden_abs.err || { tail -5 $OUT/garden_abs.err; exit 1; }
timeout -k 10 200 python3 $R/tools/raster_ab.py --libs $LIBS --scene wolf --absgrad --seg $SEG --reps 20 > $OUT/wolf_abs_seg.json 2> $OUT/wolf_abs_seg.err || { tail -5 $OUT/wolf_abs_seg.err; exit 1; }
timeout -k 10 200 python3 $R/tools/raster_ab.py --libs $LIBS --scene lego --seg $SEG --reps 20 > $OUT/lego_seg.json 2> $OUT/lego_seg.err || { tail -5 $OUT/lego_seg.err; exit 1; }
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    print(os.path.basename(f))
    txt=open(f).read()
    try:
        d=json.loads(txt[txt.index("{"):]) if txt.lstrip().startswith("{") else json.loads(txt)
    except Exception as e:
        print("  unparsed", e); continue
    rs = d["results"] if isinstance(d, dict) and "results" in d else d
    for r in rs:
        print("  %-28s bwd %7.1f us (min %7.1f)  rel diff vs first %.2e" % (os.path.basename(r["lib"]), r["bwd_us_median"], r["bwd_us_min"], r["rel_diff_vs_first"]))
PY
