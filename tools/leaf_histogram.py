"""Leaf-size histogram of every mesh the benchmark scenes use, as the reference's builder
(BVH::initialize(4), src/bvh.rs:52-178) leaves them: the assets, the procedural knot that
stands in for the absent dragon, and the subdivided assets.  CPU only.
Usage: python tools/leaf_histogram.py > profiles/r02_leaf_histograms.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host  # noqa: E402

BINS = [1, 2, 3, 4, 5, 8, 16, 24, 64, 256, 1024, 1 << 30]


def report(name, nodes):
    leaves = nodes[nodes["is_leaf"] == 1]
    sizes = (leaves["end"] - leaves["start"]).astype(np.int64)
    cells = []
    for lo, hi in zip(BINS[:-1], BINS[1:]):
        n = int(((sizes >= lo) & (sizes < hi)).sum())
        label = f"{lo}" if hi == lo + 1 else (f"{lo}+" if hi == 1 << 30 else f"{lo}-{hi - 1}")
        cells.append(f"{label}:{n}")
    big = sizes[sizes >= 24]
    print(f"{name:<34} triangles {int(sizes.sum()):>8}  nodes {len(nodes):>8}  leaves {len(leaves):>7}  "
          f"largest {int(sizes.max()):>5}  in leaves >= 24: {len(big)} leaves, {int(big.sum())} triangles")
    print(f"{'':<34} leaves by size  " + "  ".join(cells))
    print(f"{'':<34} eight largest   {sorted(sizes.tolist())[-8:]}")


s = host.Scene()
d = host.default_asset_dir()
for n in ("suzanne", "teapot"):
    report(f"{n} (asset)", s.mesh(s.load_model_file(f"{d}/{n}.mirtmesh"))[3])
report("knot 2722x160 seed 1 (config 3)", s.mesh(s.load_knot(2722, 160, 1))[3])
report("suzanne x4^5, jittered", s.mesh(s.load_subdivided("suzanne", 5))[3])
report("teapot x4^4, jittered", s.mesh(s.load_subdivided("teapot", 4))[3])
report("teapot x4^3, jittered", s.mesh(s.load_subdivided("teapot", 3))[3])
report("knot 1361x80 seed 1", s.mesh(s.load_knot(1361, 80, 1))[3])
report("knot 2722x160 seed 2", s.mesh(s.load_knot(2722, 160, 2))[3])
