"""Phase breakdown of the extend kernel from a -DMIRT_PHASE_TIMING build
(MIRT_DEVICE_LIB=libmirt_phase.so python tools/phase_breakdown.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, pathtracer
arrays, info = host.build_config_scene(3)
W, H = 1920, 1080
src = host.make_sample_sources(W, H, info.seed); cam = host.default_camera(W, H)
pt = pathtracer.Pathtracer(0); pt.upload_scene(arrays)
pt.create_state(W, H, W * H, src, pathtracer.default_params(max_bounces=8, seed=info.seed)); pt.set_camera(cam)
pt.render_spp(2); pt.stats_reset(); pt.set_event_timing(True)
t = time.time(); pt.render_spp(4); dt = time.time() - t; st = pt.stats()
ph = list(st.engine_phase)
print(f"{st.extend_rays/dt/1e6:.1f} Mrays/s, extend {st.extend_ms/st.extend_launches:.3f} ms/launch x {st.extend_launches}")
if ph[4]:
    names = ["refill", "node", "small-leaf", "coop-leaf"]
    tot = ph[4]
    for n, v in zip(names, ph[:4]):
        print(f"  {n:<11} {100*v/tot:5.1f} % of wave cycles")
    print(f"  node-phase iterations {ph[5]:.3e} ({ph[1]/max(ph[5],1):.0f} cyc each); coop iterations {ph[6]/64:.3e} wave-its ({ph[3]/max(ph[6]/64,1):.0f} cyc each); rounds {ph[7]:.3e}")
    print(f"  rays {st.extend_rays}: node its per ray-wave {ph[5]*64/st.extend_rays:.1f}")
