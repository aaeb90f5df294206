"""Soak run (not collected): many seeds of ray sets, comb scenes with random leaf-size mixes and short renders,
device against oracle, bit for bit.  Usage on the GPU box: python tests/debug/wip_soak.py [rounds]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from raytracer_amd import ffi, host, pathtracer  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from conftest import hits_equal, random_rays  # noqa: E402
from test_gpu_parity import _comb_scene  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t0 = time.time()
bad = 0
with pathtracer.Pathtracer(0) as pt:
    for r in range(rounds):
        rng = np.random.default_rng(1000 + r)
        # 1. comb scenes: leaf sizes drawn from every class (1..4, 5..23, 24..63, 64..300)
        sizes = [int(x) for x in np.concatenate([rng.integers(1, 5, 12), rng.integers(5, 24, 10), rng.integers(24, 64, 3),
                                                 rng.integers(64, 300, 2)])]
        rng.shuffle(sizes)
        arrays = _comb_scene(sizes, 100 + r)
        sc = po.Scene(arrays)
        pt.upload_scene(arrays)
        rays = random_rays(20000, 500 + r, centre=(0.0, 0.0, 3.0), spread=2.0)
        # some rays leave from a triangle of the scene: exclusion inside leaves of every size
        n_tri = int(sum(sizes))
        rays["last_inst"][::3] = 0
        rays["last_prim"][::3] = rng.integers(0, n_tri, len(rays[::3]))
        want, wc = sc.trace_closest(rays, True)
        got, gc = pt.trace_closest(rays, True)
        ok = hits_equal(got, want) and gc.as_dict() == wc.as_dict()
        tmax = rng.uniform(0.5, 6.0, len(rays)).astype("<f4")
        wa, wca = sc.trace_any(rays, tmax, True)
        ga, gca = pt.trace_any(rays, tmax, True)
        ok = ok and hits_equal(ga, wa) and gca.as_dict() == wca.as_dict()
        W, H = 48, 32
        src = host.make_sample_sources(W, H, r)
        cam = host.default_camera(W, H)
        prm = dict(max_bounces=5, seed=r)
        pt.create_state(W, H, 700 + 13 * r, src, pathtracer.default_params(**prm))
        pt.set_camera(cam)
        pt.render_spp(2)
        s4, cnt = pt.read_accum()
        w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(**prm), 0, 2)
        ok = ok and np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
        print(f"round {r}: comb {len(sizes)} leaves {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        # 2. the benchmark configurations at small size, both shading paths
        for cfg in (1, 2, 4, 5):
            kw = dict(dragon_n_u=40, dragon_n_v=10) if cfg >= 3 else {}
            arrays, info = host.build_config_scene(cfg, **kw)
            sc = po.Scene(arrays)
            pt.upload_scene(arrays)
            centre, spread = ((0.0, 0.0, 14.0), 10.0) if cfg == 4 else ((0.0, 0.0, 3.0), 2.0)
            rays = random_rays(15000, 900 + 10 * r + cfg, centre=centre, spread=spread)
            want, wc = sc.trace_closest(rays, True)
            got, gc = pt.trace_closest(rays, True)
            ok = hits_equal(got, want) and gc.as_dict() == wc.as_dict()
            W, H = 40, 30
            src = host.make_sample_sources(W, H, info.seed + r)
            cam = host.default_camera(W, H)
            # uber; per-material without light sampling; with it, in one kernel and in a kernel per material
            for mode, nee, stage in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
                prm = dict(max_bounces=4, seed=info.seed + r, shade_mode=mode, nee=nee)
                pt.create_state(W, H, 900, src, pathtracer.default_params(stage_kernels=stage, **prm))
                pt.set_camera(cam)
                pt.render_spp(2)
                s4, cnt = pt.read_accum()
                w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(**prm), 0, 2)
                ok = ok and np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
            print(f"round {r}: config {cfg} {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
print(f"{bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
