"""Cost of a full scene upload against an instance-only update (configuration 3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer
arrays, info = host.build_config_scene(3)
pt = pathtracer.Pathtracer(0)
for name, f in (("mirt_scene_upload", pt.upload_scene), ("mirt_scene_update_instances", pt.update_instances)):
    f(arrays)
    t0 = time.perf_counter()
    for _ in range(5):
        f(arrays)
    print(f"{name:<30} {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms")
