"""Latency of ONE long ray walking alone: a steady-state extension queue of the per-frame call (config 3,
512 x 512, one slot per pixel) is dumped from the device, the oracle counts every ray's node visits, and the
longest rays are traced alone (a launch of one ray: one lane of one wave on an otherwise idle GPU) and in
small groups.  ms per launch / visits = what a dependent step of the walk costs with nothing else running.
Runs on the GPU box: [MIRT_DEVICE_LIB=...] python tools/long_ray_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, pathtracer
from oracle import pyoracle as po

cache = "gpurun_out/long_rays.npz" if os.path.isdir("gpurun_out") else "/tmp/long_rays.npz"
arrays, info = host.build_config_scene(3)
pt = pathtracer.Pathtracer(0); pt.upload_scene(arrays)
W = H = 512
if os.path.exists(cache):
    z = np.load(cache); rays, visits, tris = z["rays"].view(pathtracer.ffi.Ray).reshape(-1), z["visits"], z["tris"]
else:
    src = host.make_sample_sources(W, H, info.seed, shuffle=2); cam = host.default_camera(W, H)
    pt.create_state(W, H, W * H, src, pathtracer.default_params(max_bounces=8, seed=info.seed)); pt.set_camera(cam)
    for _ in range(40): pt.step(1)
    pt.synchronize()
    allrays = pt.dump_extension_rays(W * H)
    sc = po.Scene(arrays)
    n = min(len(allrays), int(os.environ.get("SAMPLE", "80000")))
    pick = np.random.default_rng(1).choice(len(allrays), n, replace=False)
    visits = np.zeros(n, np.int64); tris = np.zeros(n, np.int64)
    t0 = time.time()
    for k, i in enumerate(pick):
        _, c = sc.trace_closest(allrays[i:i + 1], want_counts=True)
        visits[k] = c.tlas_nodes + c.blas_nodes; tris[k] = c.triangles
    order = np.argsort(-visits)[:256]
    rays, visits, tris = allrays[pick[order]].copy(), visits[order], tris[order]
    print(f"oracle counted {n} rays in {time.time() - t0:.1f} s", flush=True)
    np.savez(cache, rays=rays.view(np.uint8), visits=visits, tris=tris)
print("longest rays: node visits", visits[:8].tolist(), "triangle tests", tris[:8].tolist())
for k in (0, 1, 2, 5):
    ms = pt.bench_trace_closest(rays[k:k + 1], repeats=20)
    print(f"ray {k} alone: {ms * 1e3:.1f} us per launch, {visits[k]} node visits, {tris[k]} triangle tests -> {ms * 1e3 / visits[k]:.3f} us per node visit")
empty = pt.bench_trace_closest(rays[255:256], repeats=20)
print(f"(ray 255, {visits[255]} visits: {empty * 1e3:.1f} us per launch)")
for n in (4, 16, 64, 256):
    ms = pt.bench_trace_closest(rays[:n], repeats=20)
    print(f"the {n} longest together (one chunk, lanes of {max(1, n // 64)} wave(s)): {ms * 1e3:.1f} us per launch")
