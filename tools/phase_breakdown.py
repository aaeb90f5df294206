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
PPP = int(os.environ.get('PPP', '8'))
pt.create_state(W, H, W * H * PPP, src, pathtracer.default_params(max_bounces=8, seed=info.seed)); pt.set_camera(cam)
pt.render_spp(2); pt.stats_reset(); pt.set_event_timing(True)
t = time.time(); pt.render_spp(4 * PPP); dt = time.time() - t; st = pt.stats()
ph = list(st.engine_phase)
print(f"{st.extend_rays/dt/1e6:.1f} Mrays/s, extend {st.extend_ms/st.extend_launches:.3f} ms/launch x {st.extend_launches}")
if ph[5]:
    its = ph[5]
    rw = st.extend_rays / 64.0
    print(f"  iterations {its:.3e} ({its/rw:.1f} per 64 rays), service points {ph[0]:.3e} ({ph[0]/rw:.1f} per 64 rays), whole-wave leaves {ph[3]:.3e}")
    print(f"  lanes per iteration: walking {ph[6]/its:.1f}, node steps {ph[1]/its:.1f}, triangle steps {ph[2]/its:.1f}")
    print(f"  per ray: node steps {ph[1]/st.extend_rays:.1f}, lane triangle steps {ph[2]/st.extend_rays:.1f}")
    print(f"  wave time in whole-wave leaf scans: {100.0 * ph[7] / ph[4]:.1f} % ({ph[7] / max(ph[3], 1):.0f} cycles per scan; wave cycles counted at 100 MHz)")
