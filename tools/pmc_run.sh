#!/bin/bash
# Collects rocprofv3 PMC counters for the bench kernels in separate passes (gpurun forbids mixing
# --pmc with tracing domains).  Usage: tools/pmc_run.sh <outdir> [bench args...]
set -e
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
            "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-timing --frames-in-flight 1 --no-verify "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$REPO" "$@" <<'PY'
import csv, glob, sys, collections, os
out, repo, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, repo)
import bench
cfgname = bargs[bargs.index("--config") + 1] if "--config" in bargs else "B"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if not (k.startswith("gs_") or k.startswith("void gs_")):
            continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/pmc_summary.csv", "w") as w:
    names = sorted({c for k in agg for c in agg[k]})
    w.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(agg):
        n = max(len(v) for v in agg[k].values())
        w.write(k.replace(",", ";") + "," + str(n) + "," + ",".join("%.6g" % (sum(agg[k][c]) / len(agg[k][c])) if c in agg[k] else "" for c in names) + "\n")
import json
js = {"workload": bench.CONFIGS[cfgname]["name"], "bench_args": " ".join(bargs),
      "note": "mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them (tools/pmc_run.sh, 4 separate --pmc passes)",
      "kernels": {k.replace("void ", "").replace("<false>", "").replace("<true>", "_exact"): {c: sum(v) / len(v) for c, v in agg[k].items() if c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")} for k in agg}}
json.dump(js, open(out + "/pmc.json", "w"), indent=1)
print(open(out + "/pmc_summary.csv").read())
PY
