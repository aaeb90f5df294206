"""Runs on the GPU box: does the order of the rays in an extension queue matter?  A steady-state queue of the
headline scene (all bounces mixed) is dumped from the device and traced as it is, bucketed by direction octant,
and bucketed by octant and origin cell; prints Mrays/s of each."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer  # noqa: E402

W, H, PPP = 1920, 1080, 8
arrays, info = host.build_config_scene(3)
src = host.make_sample_sources(W, H, info.seed, shuffle=2)
cam = host.default_camera(W, H)
with pathtracer.Pathtracer(0) as pt:
    pt.upload_scene(arrays)
    pt.create_state(W, H, W * H * PPP, src, pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed))
    pt.set_camera(cam)
    pt.step(12)  # past the first generation: every bounce depth is present
    rays = pt.dump_extension_rays(W * H * PPP)
    print(len(rays), "rays in the queue")
    d, o = rays["dir"], rays["pos"]
    octant = (d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2)
    lo, hi = o.min(0), o.max(0)
    cell = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * 16).astype(np.uint32), 0, 15)
    cellkey = cell[:, 0] | (cell[:, 1] << 4) | (cell[:, 2] << 8)
    orders = {
        "as queued": np.arange(len(rays)),
        "shuffled": np.random.default_rng(1).permutation(len(rays)),
        "by octant": np.argsort(octant, kind="stable"),
        "by origin cell (16^3)": np.argsort(cellkey, kind="stable"),
        "by octant, then origin cell": np.argsort(octant * 4096 + cellkey, kind="stable"),
        "by origin cell, then octant": np.argsort(cellkey * 8 + octant, kind="stable"),
    }
    for name, order in orders.items():
        ms = pt.bench_trace_closest(rays[order], repeats=5)
        print(f"{name:32s} {ms:8.3f} ms  {len(rays) / ms / 1e3:8.1f} Mrays/s")
