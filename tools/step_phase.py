"""Where a frame of mirt_step goes at the reference's operating point (512 x 512, one path slot per
pixel), from a -DMIRT_PHASE_TIMING build: MIRT_DEVICE_LIB=libmirt_phase.so python tools/step_phase.py
Times are the 100 MHz constant clock, summed over the waves of k_extend unless said otherwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer
W = int(os.environ.get("W", "512")); H = int(os.environ.get("H", "512")); PPP = int(os.environ.get("PPP", "1"))
F = int(os.environ.get("FRAMES", "64"))
arrays, info = host.build_config_scene(3)
src = host.make_sample_sources(W, H, info.seed, shuffle=2); cam = host.default_camera(W, H)
pt = pathtracer.Pathtracer(0); pt.upload_scene(arrays)
pt.create_state(W, H, W * H * PPP, src, pathtracer.default_params(max_bounces=8, seed=info.seed)); pt.set_camera(cam)
for _ in range(32): pt.step(1)
pt.synchronize(); pt.stats_reset(); pt.set_event_timing(True)
slowest = []
for _ in range(F):
    pt.step(1)
pt.synchronize()
st = pt.stats(); ph = [int(x) for x in st.engine_phase]
n = int(st.extend_launches); rays = int(st.extend_rays)
us = lambda ticks: ticks / 100.0
waves = (W * H * PPP + 63) // 64
print(f"{W}x{H} x{PPP}: k_extend {st.extend_ms / n:.3f} ms per frame, {rays / n:.0f} rays per frame, {rays / (st.extend_ms * 1e3):.1f} Mrays/s in the kernel")
print(f"  slowest wave of any frame: {us(ph[8]):.0f} us (kernel {st.extend_ms / n * 1e3:.0f} us)")
tot = ph[4]
print(f"  wave time summed: {us(tot) / n / 1e3:.1f} ms per frame over all waves = {us(tot) / n / min(waves, 4096):.0f} us per wave")
print(f"  step loop {100 * ph[15] / tot:.1f} %, service points {100 * ph[14] / tot:.1f} %, whole-wave scans {100 * ph[7] / tot:.1f} % (inside service)")
print(f"  with <= 4 lanes walking: step loop {100 * ph[10] / tot:.1f} %, service {100 * ph[11] / tot:.1f} %; 5..16: step loop {100 * ph[13] / tot:.1f} %, service {100 * ph[2] / tot:.1f} %")
its = ph[5]
print(f"  step iterations {its / n:.3e} per frame: {100 * ph[9] / its:.1f} % with <= 4 walking, {100 * ph[12] / its:.1f} % with 5..16; lanes walking per iteration {ph[6] / its:.1f}, on a node {ph[1] / its:.1f}")
if ph[9]: print(f"  per step iteration with <= 4 walking: {us(ph[10]) / ph[9]:.3f} us; service points {ph[0] / n:.3e} per frame, {us(ph[14]) / max(ph[0], 1):.3f} us each")
if ph[12]: print(f"  per step iteration with 5..16 walking: {us(ph[13]) / ph[12]:.3f} us")
rest = its - ph[9] - ph[12]
if rest: print(f"  per step iteration with > 16 walking: {us(ph[15] - ph[10] - ph[13]) / rest:.3f} us")
# per-wave sums of the LAST frame (parked in the convergence flags by the diagnostic build): the slowest waves
import numpy as np
fl = pt.read_flags()
n_waves = 256 * 16
if len(fl) >= 16 * n_waves:
    w = fl[:16 * n_waves].reshape(n_waves, 16).astype(np.int64)
    order = np.argsort(-w[:, 4])
    tick = w[:, 4].max() / (st.extend_ms / n * 1e3)  # ticks per microsecond, taking the slowest wave for the kernel's time
    print(f"  last frame, per wave (ticks per us ~ {tick:.0f}): busy waves {int((w[:, 5] > 0).sum())}, median wave {np.median(w[w[:, 5] > 0, 4]) / tick:.0f} us, slowest {w[order[0], 4] / tick:.0f} us")
    for k in order[:6]:
        r = w[k]
        print(f"    wave {k}: {r[4] / tick:.0f} us = step loop {r[15] / tick:.0f} + service {r[14] / tick:.0f} (scans {r[7] / tick:.0f}); iterations {r[5]}, of them <=4 lanes {r[9]} ({r[10] / tick:.0f} us), 5..16 lanes {r[12]} ({r[13] / tick:.0f} us); "
              f"service points {r[0]} (with <=4 lanes {r[11] / tick:.0f} us, 5..16 {r[2] / tick:.0f} us); lane node steps {r[1]}, walking-lane iterations {r[6]}")
