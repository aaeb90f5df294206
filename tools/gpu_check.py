"""First-contact GPU check: arithmetic contract, ray-set parity, render parity, timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, ffi, pathtracer
from oracle import pyoracle as po

def rand_rays(n, seed, arrays):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, ffi.Ray)
    rays["pos"] = rng.uniform(-2, 2, (n, 3)).astype("<f4") + np.array([0, 0, 3], "<f4")
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["dir"] = d.astype("<f4")
    rays["last_inst"] = ffi.NO_PRIM; rays["last_prim"] = ffi.NO_PRIM
    return rays

def cmp_hits(a, b, what):
    same = (a["instance_id"] == b["instance_id"]) & (a["triangle_id"] == b["triangle_id"]) & \
           (a["t"].view("<u4") == b["t"].view("<u4")) & (a["front_face"] == b["front_face"])
    print(f"{what}: {same.sum()}/{len(a)} identical; hits {np.sum(a['instance_id'] != ffi.NO_HIT)}")
    if not same.all():
        i = np.nonzero(~same)[0][:5]; print(" gpu", a[i]); print(" cpu", b[i])
    return bool(same.all())

pt = pathtracer.Pathtracer(0)
rng = np.random.default_rng(1)
n = 100000
a = (rng.normal(size=n) * 10 ** rng.uniform(-20, 20, n)).astype("<f4")
b = (rng.normal(size=n) * 10 ** rng.uniform(-20, 20, n)).astype("<f4")
c = (rng.normal(size=n) * 10 ** rng.uniform(-20, 20, n)).astype("<f4")
g = pt.math_selftest(a, b, c).view("<u4"); o = po.math_selftest(a, b, c).view("<u4")
print("math selftest identical:", np.array_equal(g, o), "mismatches", np.sum(g != o))
if not np.array_equal(g, o):
    bad = np.nonzero(g != o)[0][:10]; print(bad % 6, g[bad], o[bad])

ok = True
for cfg in (1, 2):
    arrays, info = host.build_config_scene(cfg)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = (info.width, info.height) if cfg == 1 else (256, 144)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed)
    for bounce in (0, 1, 2):
        rays = sc.bounce_rays(cam, src, W, H, oprm, 0, bounce)
        rays = rays[rays["last_inst"] != ffi.NO_HIT]
        t = time.time(); oh, oc = sc.trace_closest(rays, True); tc = time.time() - t
        gh, gc = pt.trace_closest(rays, True)
        ok &= cmp_hits(gh, oh, f"cfg{cfg} bounce{bounce} closest ({len(rays)} rays, cpu {tc:.2f}s)")
        print("   counts gpu", gc.as_dict(), "equal", gc.as_dict() == oc.as_dict()); ok &= gc.as_dict() == oc.as_dict()
    rays = rand_rays(200000, 7, arrays)
    oh = sc.trace_closest(rays); gh = pt.trace_closest(rays); ok &= cmp_hits(gh, oh, f"cfg{cfg} random closest")
    tmax = np.random.default_rng(3).uniform(0.5, 6.0, len(rays)).astype("<f4")
    oa, oc = sc.trace_any(rays, tmax, True); ga, gc = pt.trace_any(rays, tmax, True)
    ok &= cmp_hits(ga, oa, f"cfg{cfg} random any-hit"); print("   any counts equal", gc.as_dict() == oc.as_dict())
    # render parity
    pt.create_state(W, H, W * H // 2, src, prm); pt.set_camera(cam)
    spp = 2
    t = time.time(); pt.render_spp(spp); tg = time.time() - t
    gs, gcnt = pt.read_accum(); st = pt.stats()
    t = time.time(); os4, ocnt, ost = sc.render(cam, src, W, H, oprm, 0, spp); tcpu = time.time() - t
    print(f"cfg{cfg} render {W}x{H} spp{spp}: gpu {tg:.3f}s cpu {tcpu:.3f}s; sums equal {np.array_equal(gs, os4)}, counts equal {np.array_equal(gcnt, ocnt)}")
    print("   gpu stats ext", st.extend_rays, "shade", st.shade_events, "fin", st.finished_samples, "frames", st.frames, "| cpu ext", ost.extend_rays, "shade", ost.shade_events)
    ok &= np.array_equal(gs, os4) and np.array_equal(gcnt, ocnt)
    if not np.array_equal(gs, os4):
        d = np.nonzero(gs != os4)[0]; print("   diff entries", len(d), d[:10], gs[d[:10]], os4[d[:10]])
    rgb = pt.read_output(); orgb = po.resolve(os4, ocnt, src, W, H, oprm)
    print("   pixels equal", np.array_equal(rgb, orgb)); ok &= np.array_equal(rgb, orgb)
print("ALL OK" if ok else "MISMATCH")

# config 3 timing (dragon-class)
t = time.time(); arrays, info = host.build_config_scene(3); print("cfg3 build", time.time() - t, {k: len(v) for k, v in arrays.items() if len(v)})
pt.upload_scene(arrays)
W, H = 1920, 1080
src = host.make_sample_sources(W, H, info.seed); cam = host.default_camera(W, H)
prm = pathtracer.default_params(max_bounces=8, seed=info.seed)
pt.create_state(W, H, W * H, src, prm); pt.set_camera(cam)
pt.set_event_timing(1)
for spp in (1, 4):
    pt.stats_reset(); t = time.time(); pt.render_spp(spp); dt = time.time() - t; st = pt.stats()
    print(f"cfg3 1080p spp{spp}: {dt:.3f}s rays {st.extend_rays} -> {st.extend_rays/dt/1e6:.1f} Mrays/s; frames {st.frames}; extend {st.extend_ms:.1f}ms/{st.extend_launches} shade {st.shade_ms:.1f}ms sample {st.sample_ms:.1f}ms")
sc3 = po.Scene(arrays)
oprm = po.default_params(max_bounces=8, seed=info.seed)
for bounce in (0, 1, 3):
    rays = sc3.bounce_rays(cam, src[:262144], W, H, oprm, 0, bounce); rays = rays[rays["last_inst"] != ffi.NO_HIT]
    t = time.time(); oh, oc = sc3.trace_closest(rays, True); tc = time.time() - t
    gh, gc = pt.trace_closest(rays, True)
    cmp_hits(gh, oh, f"cfg3 bounce{bounce} ({len(rays)} rays cpu {tc:.2f}s)")
    ms = pt.bench_trace_closest(rays, 5)
    ab = oc.algorithmic_bytes(112)
    print(f"   counts equal {gc.as_dict() == oc.as_dict()} per ray: nodes {(oc.blas_nodes+oc.tlas_nodes)/oc.rays:.1f} tris {oc.triangles/oc.rays:.1f}; gpu {ms:.3f} ms -> {len(rays)/ms/1e3:.1f} Mrays/s, {ab/ms/1e6:.1f} GB/s algorithmic")
