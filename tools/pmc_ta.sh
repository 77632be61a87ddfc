#!/bin/bash
# Texture-addresser / vector-L1 counters of the blend kernel (one --pmc pass per group).  Usage: tools/pmc_ta.sh <outdir> [bench args]
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/t$i" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-timing --frames-in-flight 1 --no-verify "$@" > "$OUT/t$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/t$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/t*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "blend" in k or "sweep" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
