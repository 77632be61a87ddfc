#!/bin/bash
# A/B of environment switches on ONE box.  Usage: tools/ab_env.sh <tag> "<bench args>" "VAR=val" "" ...
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py --no-cpu --no-verify --steps 100 --warmup 10 $ARGS > $OUT/e$i.json 2> $OUT/e$i.err || echo "variant $i failed"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/e$i.json"))
    print("[${e:-default}]", round(d["value"], 1), "fps", {k: round(v["us"]) for k, v in d.get("stages", {}).items()})
except Exception as ex:
    print("[${e:-default}] failed", ex)
PY
done
