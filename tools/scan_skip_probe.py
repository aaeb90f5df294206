"""How often a pass of a whole-wave scan (64 triangles of a bail-out leaf) has no lane that could accept on the
u test alone, for the rays a steady-state extension queue of config 3 actually sends into the dragon stand-in's
biggest leaf.  Runs on the GPU box (the rays are dumped from the device): python tools/scan_skip_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, pathtracer
arrays, info = host.build_config_scene(3)
pt = pathtracer.Pathtracer(0); pt.upload_scene(arrays)
W = H = 512
src = host.make_sample_sources(W, H, info.seed, shuffle=2); cam = host.default_camera(W, H)
pt.create_state(W, H, W * H, src, pathtracer.default_params(max_bounces=8, seed=info.seed)); pt.set_camera(cam)
for _ in range(40): pt.step(1)
pt.synchronize()
rays = pt.dump_extension_rays(W * H)
nodes = arrays["blas_nodes"]; go = arrays["geometry_offsets"]
leafmask = nodes["is_leaf"] == 1
sizes = np.where(leafmask, nodes["end"].astype(np.int64) - nodes["start"].astype(np.int64), 0)
ni = int(np.argmax(sizes)); L = nodes[ni]; g = int(np.searchsorted(go["blas_node"], ni, side="right") - 1)
inst = [i for i in range(len(arrays["instances"])) if arrays["instances"][i]["geometry"] == g][0]
t = arrays["transforms"][arrays["instances"][inst]["transform"]]
assert not np.any(t["rotation"][:3]), "probe assumes an unrotated instance"
sc, tr = t["scale"][:3].astype(np.float64), t["translation"][:3].astype(np.float64)
V = arrays["vertices"]["position"][:, :3]; I = arrays["indices"]["v"][:, :3]
tri = I[go["index"][g] + L["start"]: go["index"][g] + L["end"]].astype(np.int64) + go["vertex"][g]
p0 = V[tri[:, 0]].astype(np.float64); e1 = V[tri[:, 1]].astype(np.float64) - p0; e2 = V[tri[:, 2]].astype(np.float64) - p0
lb, ub = L["lb"][:3].astype(np.float64), L["ub"][:3].astype(np.float64)
o = (np.stack([rays["pos"][:, 0], rays["pos"][:, 1], rays["pos"][:, 2]], 1).astype(np.float64) - tr) / sc
d = np.stack([rays["dir"][:, 0], rays["dir"][:, 1], rays["dir"][:, 2]], 1).astype(np.float64) / sc
with np.errstate(divide="ignore", invalid="ignore"):
    inv = 1.0 / d
    t0 = (lb - o) * inv; t1 = (ub - o) * inv
tmin = np.nanmax(np.minimum(t0, t1), axis=1); tmax = np.nanmin(np.maximum(t0, t1), axis=1)
reach = np.flatnonzero((tmin <= tmax) & (tmax >= 0))
print(f"leaf of {len(p0)} triangles; {len(reach)} of {len(rays)} rays cross its box")
skip = total = 0; frac = []
for k in reach[:3000]:
    q = np.cross(d[k], e2); alpha = (e1 * q).sum(1); s = o[k] - p0; dd = (s * q).sum(1)
    opposite = (dd < 0) != (alpha < 0)
    sure = (np.abs(alpha) < 1e-7) | (opposite & (np.abs(dd) >= 1e-10)) | (~opposite & (np.abs(dd) > 1.001 * np.abs(alpha)))
    frac.append(1 - sure.mean())
    for b in range(0, len(sure), 64):
        total += 1; skip += bool(sure[b:b + 64].all())
print(f"triangles not surely rejected by u: {100 * np.mean(frac):.2f} %; passes of 64 with none: {100 * skip / total:.1f} %")
