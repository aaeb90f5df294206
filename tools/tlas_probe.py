"""Runs on the GPU box: time mirt_build_tlas on a big instance set (device ms from HIP events, wall per
call with the host-side matrices and the copies) next to the host model's bind() of the same scene
(which builds the same TLAS on one core, plus the small tables), and the transforms-only update of a
resident scene."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(1)
s = host.Scene()
cube = s.load_cube()
mat = s.add_material(host.make_material())
L, h = s._L, s._h
t = host.make_transform()
for i in range(n):
    t["scale"][0, :3] = rng.uniform(0.2, 2.0, 3)
    t["rotation"][0, :3] = rng.uniform(-3, 3, 3)
    t["translation"][0, :3] = rng.uniform(-200, 200, 3)
    L.mirth_spawn(h, t.ctypes.data, cube, mat)
t0 = time.perf_counter()
arrays = s.bind()
t_bind = time.perf_counter() - t0
roots = arrays["blas_nodes"][arrays["geometry_offsets"]["blas_node"]]
aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
with pathtracer.Pathtracer(0) as pt:
    pt.build_tlas(aabbs, arrays["transforms"], arrays["instances"])
    t0 = time.perf_counter()
    nodes, ids, ms = pt.build_tlas(aabbs, arrays["transforms"], arrays["instances"])
    wall = time.perf_counter() - t0
    # transforms only, on a resident scene: nothing comes back to the host
    pt.upload_scene(arrays)
    moved = arrays["transforms"].copy()
    moved["translation"][:, :3] += rng.normal(0, 1.0, (len(moved), 3)).astype("<f4")
    pt.update_transforms(moved)
    t0 = time.perf_counter()
    ms_refit = pt.update_transforms(moved)
    wall_refit = time.perf_counter() - t0
same = nodes.tobytes() == arrays["tlas_nodes"].tobytes() and np.array_equal(ids, arrays["tlas_to_instances"])
print(f"{n} instances -> {len(nodes)} TLAS nodes: host bind() {t_bind * 1e3:.1f} ms, device build {ms:.2f} ms "
      f"({wall * 1e3:.1f} ms per call with host matrices and copies), byte-identical: {same}; "
      f"mirt_scene_update_transforms on the resident scene {ms_refit:.2f} ms on the device, {wall_refit * 1e3:.1f} ms per call")
