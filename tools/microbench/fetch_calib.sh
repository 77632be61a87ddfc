#!/bin/bash
# GPU box: FETCH_SIZE calibration (see fetch_calib.hip).  Usage: tools/microbench/fetch_calib.sh <outdir>
OUT=${1:-gpurun_out/fetch_calib}; mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
HERE=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$HERE/fetch_calib.hip" -o "$OUT/fetch_calib" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$OUT/p1" -- "$OUT/fetch_calib" > "$OUT/run.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p2" -- "$OUT/fetch_calib" > "$OUT/run2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/p2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e3
known = 2 << 30
with open(out + "/fetch_calib.txt", "w") as w:
    for k, v in sorted(agg.items()):
        line = "%-14s FETCH_SIZE %.0f KiB per launch = %.3f x the known %d bytes; %.0f us -> %.2f TB/s of known bytes" % (
            k, v[-1], v[-1] * 1024 / known, known, dur.get(k, 0), known / (dur.get(k, 1) * 1e-6) / 1e12 if k in dur else 0)
        print(line); w.write(line + "\n")
PY
