#!/bin/bash
# GPU box: the -m gpu suite (with durations) and a default bench line.  Usage: tools/gpu_suite.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s --durations=25 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -45 $OUT/pytest.log
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "frame_verified", "frame_verified_detail", "exact_blend", "capacity", "valid")})
print(d.get("cpu_baseline"))
print({k: v["us"] for k, v in d.get("stages", {}).items()}, d.get("config"))
print(d.get("one_frame_in_flight"), "frames_in_flight", d.get("frames_in_flight"), d.get("roofline"))
PY
