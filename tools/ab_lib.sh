#!/bin/bash
# A/B of library builds (tools/build_variant.py) on ONE box.  Usage: tools/ab_lib.sh <tag> "<bench args>" <lib or ""> ...
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ -n "$lib" ]; then export GSPLAT_LIB=$PWD/gaussian-splatting-wgpu_amd/lib/variants/$lib.so; else unset GSPLAT_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu --no-verify --steps 100 --warmup 10 $ARGS > $OUT/l$i.json 2> $OUT/l$i.err || echo "variant $i failed"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/l$i.json"))
    print("[${lib:-product}]", round(d["value"], 1), "fps", {k: round(v["us"]) for k, v in d.get("stages", {}).items()})
except Exception as e:
    print("[${lib:-product}] failed", e)
PY
done
