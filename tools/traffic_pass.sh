#!/bin/bash
# Runs on the GPU box: the memory-side traffic of the extend kernel from the PMC counters,
# FETCH_SIZE and WRITE_SIZE in separate passes over the default bench.py workload, plus a
# calibration of FETCH_SIZE for 32-byte gathers; writes profiles/traffic.json (copied back
# through gpurun_out/).  Usage: tools/traffic_pass.sh <git head the tree was taken at>
set -u
head=${1:-unknown}
out=gpurun_out/traffic
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
for geo in knot suzanne; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${geo}_$c" -- python3 bench.py $args --geometry $geo > "$out/bench_${geo}_$c.json" 2> "$out/pmc_${geo}_$c.err" || { echo "pmc pass $geo $c failed"; tail -3 "$out/pmc_${geo}_$c.err"; exit 1; }
  done
done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_cal" -- python3 tools/gather_calibrate.py > "$out/cal.json" 2> "$out/cal.err" || { echo "calibration pass failed"; tail -3 "$out/cal.err"; exit 1; }
python3 - "$out" "$head" <<'PY'
import csv, glob, json, os, re, sys, collections
sys.path.insert(0, os.getcwd())
import bench
out, head = sys.argv[1], sys.argv[2]
def per_kernel(d, counter):
    sums = collections.defaultdict(float); calls = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"][:40]
            sums[k] += float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
    return {k: (v, len(calls[k])) for k, v in sums.items()}
cal = per_kernel(os.path.join(out, "pmc_cal"), "FETCH_SIZE")
calj = json.loads(open(os.path.join(out, "cal.json")).read().strip().splitlines()[-1])
cal_kb, cal_n = cal["k_gather_chase"]
props_cus = 256
lane_steps = cal_n * props_cus * 1024 * calj["steps"]
calibration = {"pattern": "dependent 32-byte gathers from a 2 GiB table (beyond L2 and Infinity Cache)",
               "counter_bytes_per_lane_step": cal_kb * 1024 / lane_steps, "dispatches": cal_n,
               "lane_steps_per_s": calj["lane_steps_per_s"]}
entries = []
for geo in ("knot", "suzanne"):
    fetch = per_kernel(os.path.join(out, f"pmc_{geo}_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(out, f"pmc_{geo}_WRITE_SIZE"), "WRITE_SIZE")
    b = json.loads(open(os.path.join(out, f"bench_{geo}_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    kx = "k_extend<false>"
    f_kb, f_n = fetch[kx]; w_kb, w_n = write[kx]
    entries.append({
     "workload": b["roofline"]["traffic_key"],  # bench.workload_name(): configuration, size, samples per step, geometry
     "paths_in_flight": b["config"]["paths_in_flight"],
     "measured_at_head": head,
     "device_sources_sha": bench.device_sources_sha(),
     "fetch_size_bytes": f_kb * 1024 / f_n, "write_size_bytes": w_kb * 1024 / w_n,
     "extend_dispatches": [f_n, w_n],
     "memory_side_bytes_per_extend_launch": f_kb * 1024 / f_n + w_kb * 1024 / w_n,
     "fetch_calibration": calibration,
     "all_kernels_fetch_bytes_per_dispatch": {k: v * 1024 / n for k, (v, n) in fetch.items()},
     "all_kernels_write_bytes_per_dispatch": {k: v * 1024 / n for k, (v, n) in write.items()},
    })
res = {"entries": entries,
 "note": "FETCH_SIZE / WRITE_SIZE (KB) of k_extend<false>, separate --pmc passes, per dispatch, for the default bench.py "
         "workload on both stand-in geometries. They count the L2's "
         "memory-side requests, Infinity-Cache hits included, so this is an upper bound of HBM bytes. The gfx950 "
         "doubling applies to wide coalesced streaming reads; this kernel gathers 32-B and 48-B records, and the "
         "calibration row gives what the counter reads per cache-missing 32-byte gather."}
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps([{k: e[k] for k in ("workload", "memory_side_bytes_per_extend_launch", "device_sources_sha")} for e in entries]))
PY
find "$out" -name "*.csv" -size +4M -delete
