"""Debug probe (not collected): closest hits of the device against the oracle on a few scenes, with a summary of what differs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import host, pathtracer
from oracle import pyoracle as po
from conftest import random_rays

def probe(name, arrays, rays):
    with pathtracer.Pathtracer(0) as pt:
        pt.upload_scene(arrays)
        g = pt.trace_closest(rays)
    w = po.Scene(arrays).trace_closest(rays)
    bad = np.nonzero((g["instance_id"] != w["instance_id"]) | (g["triangle_id"] != w["triangle_id"]) | (g["t"].view("u4") != w["t"].view("u4")))[0]
    print(name, "rays", len(rays), "differ", len(bad))
    for i in bad[:8]:
        print("   ray", i, "gpu", g[i], "oracle", w[i])
    if len(bad):
        miss_g = (g["instance_id"][bad] == 0xFFFFFFFF).sum(); miss_w = (w["instance_id"][bad] == 0xFFFFFFFF).sum()
        print("   gpu misses", miss_g, "oracle misses", miss_w, "gpu t > oracle t", (g["t"][bad] > w["t"][bad]).sum())

for cfg, kw, centre, spread in ((1, {}, (0, 0, 3.0), 2.0), (3, dict(dragon_n_u=48, dragon_n_v=12), (0, 0, 3.0), 2.0), (4, dict(dragon_n_u=48, dragon_n_v=12), (0.0, 0.0, 14.0), 10.0)):
    arrays, info = host.build_config_scene(cfg, **kw)
    rays = random_rays(50000, 4, centre=centre, spread=spread)
    probe(f"config{cfg}", arrays, rays)
