#!/bin/bash
# Runs on the GPU box: one rocprofv3 PMC pass over a short bench.py run, printing per-kernel sums.
# Usage: tools/pmc_pass.sh <name> <counter> [<counter>...]
set -u
name=$1; shift
out=gpurun_out/pmc_$name
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench.json" 2> "$out/err.txt" || { echo "pass $name failed"; tail -3 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections, re
d = sys.argv[1]
sums = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"][:40]
        sums[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
for k, cs in sums.items():
    if k.startswith("k_extend"):
        print(k, "dispatches", len(calls[k]), " ".join(f"{c}={v / len(calls[k]):.4g}" for c, v in sorted(cs.items())), "(per dispatch)")
PY
find "$out" -name "*.csv" -size +4M -delete
