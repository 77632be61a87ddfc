#!/bin/bash
# A/B bench variants on ONE box.  Usage: tools/ab.sh <tag> "<args A>" "<args B>" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --no-cpu --no-verify --steps 100 --warmup 10 $a > $OUT/v$i.json 2> $OUT/v$i.err || echo "variant $i failed"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/v$i.json"))
    print("[$a]", round(d["value"], 1), "fps", {k: round(v["us"]) for k, v in d.get("stages", {}).items()}, "I", d["config"]["intersections"], "proc", d["config"]["processed"], "eval", d["config"]["block_evaluated"])
except Exception as e:
    print("[$a] failed", e)
PY
done
