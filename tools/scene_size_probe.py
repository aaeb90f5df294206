"""Lane-steps per second of the extend kernel against the size of the dragon
stand-in (configuration 3): does the working set matter?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer
W, H = 1920, 1080
for nu, nv in ((2722, 160), (1360, 80), (680, 40), (170, 10)):
    arrays, info = host.build_config_scene(3, dragon_n_u=nu, dragon_n_v=nv)
    src = host.make_sample_sources(W, H, info.seed, shuffle=2)
    pt = pathtracer.Pathtracer(0)
    pt.upload_scene(arrays)
    pt.create_state(W, H, W * H * 16, src, pathtracer.default_params(max_bounces=8, seed=info.seed))
    pt.set_camera(host.default_camera(W, H))
    pt.render_spp(16)
    pt.stats_reset(); pt.set_event_timing(True)
    pt.render_spp(64)
    st = pt.stats()
    ms = st.extend_ms
    pt.set_event_timing(False); pt.stats_reset(); pt.set_counting(True)
    pt.set_sample_range(16); pt.render_spp(64)
    c = pt.stats().extend_counts
    steps = c.tlas_nodes + c.blas_nodes
    print(f"{len(arrays['indices']):>8} triangles: {c.rays / ms / 1e3:7.1f} Mrays/s in k_extend, "
          f"{steps / c.rays:6.1f} nodes + {c.triangles / c.rays:6.1f} triangles per ray, "
          f"{steps / ms / 1e6:6.1f} G node steps/s", flush=True)
    pt.close()
