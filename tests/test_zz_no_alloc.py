"""mirt_trace_closest_static — the entry point that allocates nothing — against
mirt_trace_closest, through the C ABI: same ids, t bits, front_face and visit
counts (bit-exact), on two calls in a row, the second with a smaller ray set
(the fixed device array keeps the first call's rays and results behind it)."""
import numpy as np
import pytest

from conftest import hits_equal, random_rays
from raytracer_amd import ffi, pathtracer

pytestmark = pytest.mark.gpu


def test_static_workspace_trace_matches_the_allocating_one(config_scenes, small_knot_scene):
    with pathtracer.Pathtracer(0) as pt:
        pt.upload_scene(small_knot_scene)
        big = random_rays(100003, 5)
        small = random_rays(4099, 6)
        for rays in (big, small):  # second call: fewer rays than the array still holds
            want, wc = pt.trace_closest(rays, True)
            got, gc = pt.trace_closest_static(rays, True)
            assert hits_equal(got, want)
            assert gc.as_dict() == wc.as_dict()
            assert (want["instance_id"] != ffi.NO_HIT).sum() > len(rays) // 20
        # ... and against the CPU oracle directly, so that this entry point does not stand on the other one
        from oracle import pyoracle as po
        owant, oc = po.Scene(small_knot_scene).trace_closest(small, True)
        assert hits_equal(got, owant)
        assert gc.as_dict() == oc.as_dict()
        # without counters, and with the excluded-primitive ids the bounce rays carry
        want = pt.trace_closest(small)
        hit = want["instance_id"] != ffi.NO_HIT
        again = small.copy()
        again["last_inst"] = np.where(hit, want["instance_id"], ffi.NO_PRIM)
        again["last_prim"] = np.where(hit, want["triangle_id"], ffi.NO_PRIM)
        assert hits_equal(pt.trace_closest_static(again), pt.trace_closest(again))
        # another scene on the same context, and a second context on the same GPU: same array
        arrays, _ = config_scenes(1)
        pt.upload_scene(arrays)
        assert hits_equal(pt.trace_closest_static(small), pt.trace_closest(small))
        with pathtracer.Pathtracer(0) as other:
            other.upload_scene(arrays)
            assert hits_equal(other.trace_closest_static(big[:70000]), pt.trace_closest(big[:70000]))
        # the limit is an error, not a truncation
        assert len(pt.trace_closest_static(np.zeros(0, ffi.Ray))) == 0
        assert hits_equal(pt.trace_closest_static(random_rays(131072, 8)), pt.trace_closest(random_rays(131072, 8)))
        with pytest.raises(pathtracer.MirtError, match="at most 131072 rays"):
            pt.trace_closest_static(random_rays(131073, 9))


def test_two_contexts_on_one_gpu_take_turns_on_the_static_workspace(small_knot_scene):
    """Every context on a GPU shares that GPU's one array; calls from two threads are serialised by the library
    (a per-device lock for the whole synchronous call) instead of exchanging rays and results."""
    import threading
    sets = [random_rays(90001, 21), random_rays(60007, 22)]
    with pathtracer.Pathtracer(0) as a, pathtracer.Pathtracer(0) as b:
        want = []
        for pt, rays in zip((a, b), sets):
            pt.upload_scene(small_knot_scene)
            want.append(pt.trace_closest(rays))
        bad = []

        def worker(pt, rays, expect):
            for _ in range(6):
                if not hits_equal(pt.trace_closest_static(rays), expect):
                    bad.append(1)

        threads = [threading.Thread(target=worker, args=(pt, rays, w)) for pt, rays, w in zip((a, b), sets, want)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not bad
