#!/bin/bash
# GPU box: SQ counters of the blend kernel only (two --pmc passes).  Usage: tools/pmc_blend.sh <outdir>
OUT=$1; REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1 || true
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-include-regex "blend" --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-timing --frames-in-flight 1 --no-verify > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "blend" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg): print("%-24s %.4g  (%d dispatches)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
