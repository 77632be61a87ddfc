#!/bin/bash
# usage: tools/sweep.sh "<bench args>" v1 v2 ...   -> runs bench.py with --blend-ablation v for each v
# (most ablation bits act only in the PROFILING build: python gaussian-splatting-wgpu_amd/csrc/build.py --profiling; used here when present)
PROF=$(dirname "$0")/../gaussian-splatting-wgpu_amd/lib/libgsplat_hip_prof.so
[ -f "$PROF" ] && export GSPLAT_LIB=$(cd "$(dirname "$PROF")" && pwd)/libgsplat_hip_prof.so
ARGS=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu $ARGS --blend-ablation $v > gpurun_out/sweep.json 2>/dev/null || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep.json"))
print(sys.argv[1], round(d["value"], 1), {k: v["us"] for k, v in d["stages"].items()})
PY
done
