#!/bin/bash
# One --pmc pass with the SQ counters of the blend kernel only.  Usage: tools/pmc_quick.sh <outdir> [bench args...]
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/p1" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-timing --frames-in-flight 1 --no-verify "$@" > "$OUT/p1.log" 2>&1 || echo "pass 1 failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/p2" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-timing --frames-in-flight 1 --no-verify "$@" > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "blend" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
