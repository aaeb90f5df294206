import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"  # present only in the build container; never needed by -m gpu tests


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


from raytracer_amd import ffi, host  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


@pytest.fixture(scope="session")
def config_scenes():
    """Flat scene buffers of the benchmark configurations, built once per session."""
    cache = {}

    def get(cfg, **kw):
        key = (cfg, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = host.build_config_scene(cfg, **kw)
        return cache[key]

    return get


@pytest.fixture(scope="session")
def small_knot_scene():
    """A Cornell box around a small procedural knot: has leaves larger than the
    builder's threshold of four (the one-sided-partition bail-out, up to ten
    triangles here) next to ordinary ones."""
    s = host.Scene()
    mesh = s.load_knot(96, 24, 7)
    mat = s.add_material(host.make_material(colour=(1.0, 1.0, 1.0, 1.0), metallic=0.2, roughness=0.1))
    s.spawn_cornell()
    s.spawn(host.make_transform(translation=(0.0, -0.89, 2.75)), mat, mesh)
    arrays = s.bind()
    s.close()
    return arrays


def random_rays(n, seed, centre=(0.0, 0.0, 3.0), spread=2.0):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, ffi.Ray)
    rays["pos"] = (rng.uniform(-spread, spread, (n, 3)) + np.asarray(centre)).astype("<f4")
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["dir"] = d.astype("<f4")
    rays["last_inst"] = ffi.NO_PRIM
    rays["last_prim"] = ffi.NO_PRIM
    return rays


def hits_equal(a, b):
    return bool(np.array_equal(a["instance_id"], b["instance_id"]) and np.array_equal(a["triangle_id"], b["triangle_id"])
                and np.array_equal(a["t"].view("<u4"), b["t"].view("<u4"))
                and np.array_equal(a["front_face"], b["front_face"]))
