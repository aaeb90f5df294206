"""BLAS build of the dragon-class mesh (871 040 triangles): the host builder against
mirt_build_blas on the device (device time from HIP events, and wall time of the call
including the copies of positions, faces and nodes over PCIe)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, pathtracer

s = host.Scene()
mid = s.load_knot(2722, 160, 1)
pos, _, built_faces, built_nodes = s.mesh(mid)
s.close()
# the order the faces had before the build: any order gives a valid timing input
rng = np.random.default_rng(1)
faces0 = built_faces[rng.permutation(len(built_faces))]
t0 = time.perf_counter(); hf, hn = host.build_blas(pos, faces0); t_host = time.perf_counter() - t0
pt = pathtracer.Pathtracer(0)
pt.build_blas(pos, faces0)  # warm-up (module load)
# the C call on buffers that exist already, as a host that owns its Vecs makes it
import ctypes as C
from raytracer_amd import ffi
nodes = np.zeros(2 * len(faces0), ffi.BVHNode)
best_wall, best_dev = 1e9, 1e9
for _ in range(5):
    df = faces0.copy()
    n, ms = C.c_uint32(), C.c_float()
    t0 = time.perf_counter()
    rc = pt._L.mirt_build_blas(pt._h, pos.ctypes.data, len(pos), df.ctypes.data, len(df), nodes.ctypes.data, len(nodes),
                               C.byref(n), C.byref(ms))
    w = time.perf_counter() - t0
    assert rc == 0
    best_wall, best_dev = min(best_wall, w), min(best_dev, ms.value)
dn = nodes[:n.value]
same = np.array_equal(df, hf) and dn.tobytes() == hn.tobytes()
print(f"faces {len(faces0)} nodes {len(hn)} identical {same}")
print(f"host builder {t_host * 1e3:.1f} ms; device build {best_dev:.2f} ms on the device, {best_wall * 1e3:.1f} ms per call with copies")
pt.close()
