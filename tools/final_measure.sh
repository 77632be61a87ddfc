#!/bin/bash
# The measurement set behind profiles/: default bench line (with CPU baseline), rocprofv3 kernel stats of the same
# command, PMC passes (tools/pmc_run.sh).  Usage (GPU box): tools/final_measure.sh gpurun_out/final
set -e
OUT=$1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd "$REPO"
timeout -k 10 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline'], d['cpu_baseline'], d.get('one_frame_in_flight'))"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 64 --warmup 5 --no-cpu --frames-in-flight 1 --no-verify > "$OUT/trace.log" 2>&1)
echo "trace done"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name "*kernel_trace.csv" -delete
"$REPO/tools/pmc_run.sh" "$OUT/pmc" > "$OUT/pmc.log" 2>&1
echo "pmc done"; tail -12 "$OUT/pmc.log"
