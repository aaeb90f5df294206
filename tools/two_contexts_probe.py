"""Runs on the GPU box: the headline workload as ONE context with P paths per pixel against a group of K
contexts on the same GPU with P / K each (their kernels on K streams, free to overlap): Mrays/s of both."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer  # noqa: E402

W, H, SPP = 1920, 1080, 64
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
arrays, info = host.build_config_scene(3)
src = host.make_sample_sources(W, H, info.seed, shuffle=2)
cam = host.default_camera(W, H)
prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)


def rays_of(members):
    return sum(m.stats().extend_rays + m.stats().connect_rays for m in members)


for k in (1, 2, 3):
    with pathtracer.PathtracerGroup([0] * k) as grp:
        grp.upload_scene(arrays)
        grp.create_state(W, H, W * H * P // k, src, prm)
        grp.set_camera(cam)
        grp.render_spp(SPP)  # warm-up
        for m in grp.members:
            m.synchronize()
            m.stats_reset()
        t0 = time.perf_counter()
        for _ in range(2):
            grp.render_spp(SPP)
        for m in grp.members:
            m.synchronize()
        dt = time.perf_counter() - t0
        print(f"{k} context(s), {P // k} paths per pixel each: {rays_of(grp.members) / dt / 1e6:8.1f} Mrays/s  {dt / 2 * 1e3:7.1f} ms per {SPP} spp")
