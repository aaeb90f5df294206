#!/bin/bash
# Runs on the GPU box: the path records as 48-byte records (libmirt.so) against three planes of 16 bytes
# (libmirt_planes.so, tools/build_variant.sh planes -DMIRT_PATH_PLANES=1) on the headline workload — launch times by
# HIP events and, in separate --pmc passes, the L1's miss-queue stalls and the memory-side bytes of the shade kernel and
# of the walk.  Writes gpurun_out/shade_layout_ab.txt.
set -u
out=gpurun_out/shade_layout_ab
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
for v in libmirt.so libmirt_planes.so; do
  export MIRT_DEVICE_LIB=$v
  timeout -k 10 200 python3 bench.py $args > "$out/bench_$v.json" 2> "$out/bench_$v.err" || { echo "$v: bench failed"; exit 1; }
  for c in TCP_PENDING_STALL_CYCLES_sum FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${v}_$c" -- python3 bench.py $args > "$out/pmc_${v}_$c.json" 2> "$out/pmc_${v}_$c.err" || { echo "$v $c: pmc pass failed"; exit 1; }
  done
done
python3 - "$out" > gpurun_out/shade_layout_ab.txt <<'PY'
import csv, glob, json, collections, sys
out = sys.argv[1]
print("path records: 48-byte records (libmirt.so) against three float4 planes (libmirt_planes.so, -DMIRT_PATH_PLANES=1)")
print("headline workload, 256 spp per step, 2 timed steps; counters per dispatch, separate --pmc passes\n")
for v in ("libmirt.so", "libmirt_planes.so"):
    b = json.load(open(f"{out}/bench_{v}.json"))
    n = b["roofline"]["launches"]; k = b["kernel_ms"]
    print(f"{v}: {b['value']:.1f} Mrays/s; per launch: k_extend {k['extend'] / n:.3f} ms, k_finish_shade_uber {k['shade'] / n:.3f} ms ({n} launches)")
    for c in ("TCP_PENDING_STALL_CYCLES_sum", "FETCH_SIZE", "WRITE_SIZE"):
        tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
        for f in glob.glob(f"{out}/pmc_{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for kern in ("k_finish_shade_uber<true>", "k_extend<false>"):
                    if kern in r["Kernel_Name"]:
                        tot[kern] += float(r["Counter_Value"]); cnt[kern].add(r["Dispatch_Id"])
        for kern in ("k_finish_shade_uber<true>", "k_extend<false>"):
            if cnt[kern]:
                per = tot[kern] / len(cnt[kern])
                unit = " KB" if c.endswith("SIZE") else ""
                print(f"    {c:30s} {kern:28s} {per:14.4g}{unit} per dispatch ({len(cnt[kern])} dispatches)")
    print()
PY
cat gpurun_out/shade_layout_ab.txt
find "$out" -name "*.csv" -size +4M -delete
