#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
timeout -k 10 200 python tools/visible_union.py 6m 8 > $out/visible_union_6m.json 2> $out/visible_union.err; cat $out/visible_union_6m.json | cut -c1-600
timeout -k 10 200 python tools/visible_union.py garden 8 > $out/visible_union_garden.json 2>> $out/visible_union.err; cat $out/visible_union_garden.json | cut -c1-600
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --mode scene-shard --rehearse --scene garden --steps 8 --warmup 2 --no-cpu-baseline --no-stage-profile > $out/shard_rehearsal_s2.json 2> $out/shard_rehearsal_s2.err; tail -n 2 $out/shard_rehearsal_s2.err | cut -c1-300; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r03n/shard_rehearsal_s2.json"))
    print({k:d[k] for k in ("value","unit","n_gpus","ms_per_step","scaling","async_errors") if k in d}, d.get("config"))
except Exception as e:
    print("no json", e)
PY
