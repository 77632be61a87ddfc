#!/bin/bash
# One-GPU rehearsal of bench.py's N>1 flow (gloo, every rank on GPU 0).  Usage: tools/mp_rehearse.sh <tag> <N> [bench args]
TAG=$1; N=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
  bench.py --gpus $N --steps 60 --warmup 6 --backend gloo --single-device "$@" > $OUT/n$N.json 2> $OUT/n$N.err
echo "N=$N rc=$?"
tail -3 $OUT/n$N.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/n$N.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "frames_in_flight", "slab_frame_equals_single_gpu_frame", "valid")})
    print(d["config"]["parallelism"]); print(d.get("one_frame_in_flight")); print({k: v["us"] for k, v in d.get("stages", {}).items()})
except Exception as e:
    print("no json:", e)
PY
