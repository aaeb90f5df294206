"""Runs on the GPU box under `rocprofv3 --pmc FETCH_SIZE`: one launch set of the dependent-gather
kernel on a table far larger than L2 + Infinity Cache, so that (nearly) every step is one
memory-side request of known useful size.  Prints the lane-steps made, for the calibration of
the counter in this access pattern (counter bytes per 32-byte gather that misses the caches)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import pathtracer  # noqa: E402

TABLE = 2 << 30
STEPS, REPEATS = 256, 2
with pathtracer.Pathtracer(0) as pt:
    rate = pt.bench_gather(TABLE, STEPS, REPEATS)
    # launches of k_gather_chase: one warm-up + REPEATS, each CUs x 1024 lanes x STEPS steps
    print(json.dumps({"table_bytes": TABLE, "steps": STEPS, "launches": REPEATS + 1, "lane_steps_per_s": rate}))
