"""ctypes face of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Product code (raytracer_amd/) never imports it.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class RenderStats(C.Structure):
    _fields_ = [("extend_rays", C.c_uint64), ("connect_rays", C.c_uint64), ("shade_events", C.c_uint64),
                ("finished_samples", C.c_uint64), ("extend_counts", ffi.TraversalCounts),
                ("connect_counts", ffi.TraversalCounts), ("overflowed_sums", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make oracle`")
        L = C.CDLL(path)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.oracle_scene_create.restype = vp
        L.oracle_scene_create.argtypes = [C.POINTER(ffi.SceneDesc)]
        L.oracle_scene_free.argtypes = [vp]
        L.oracle_rng_next.restype = C.c_float
        L.oracle_rng_next.argtypes = [vp]
        L.oracle_rng_init.argtypes = [u32, u32, u32, vp]
        L.oracle_trace_closest.argtypes = [vp, vp, sz, vp, C.POINTER(ffi.TraversalCounts)]
        L.oracle_trace_any.argtypes = [vp, vp, vp, sz, vp, C.POINTER(ffi.TraversalCounts)]
        L.oracle_trace_brute.argtypes = [vp, vp, sz, vp]
        L.oracle_render.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, u32, vp, vp,
                                    C.POINTER(RenderStats), C.c_int]
        L.oracle_render_wavefront.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, u32, u32,
                                              vp, vp, C.POINTER(RenderStats), C.POINTER(C.c_uint64)]
        L.oracle_step_wavefront.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, C.c_uint64, u32,
                                            vp, vp, C.POINTER(RenderStats)]
        L.oracle_render_active.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, u32, vp, u32, vp, vp, vp,
                                           C.POINTER(RenderStats), C.c_int]
        L.oracle_adapt_update.restype = u32
        L.oracle_adapt_update.argtypes = [vp, vp, vp, u32, C.c_float, u32]
        L.oracle_wavefront_per_slot.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), vp, C.c_uint64,
                                                C.c_uint64, u32, vp, vp, C.POINTER(RenderStats)]
        L.oracle_bounce_rays.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, u32, vp]
        L.oracle_connect_rays.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), u32, u32, vp, vp]
        L.oracle_resolve.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(ffi.Params), vp]
        L.oracle_math_selftest.argtypes = [vp, vp, vp, sz, vp]
        _lib = L
    return _lib


def default_params(**kw):
    """The reference's literals (MirtParams); keyword arguments override."""
    p = ffi.Params(max_bounces=128, sky_radiance=10.0, fixed_point_scale=1000.0,
                   exposure=float(np.exp2(np.float32(-2.5))), shade_mode=0, rng_mode=0, seed=0x5EED0000, nee=0,
                   adaptive=0, stage_kernels=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def rng_sequence(state, n):
    """n draws from random.slang's generator; returns (values, final state)."""
    st = np.array(state, dtype="<u4")
    vals = np.zeros(n, "<f4")
    for i in range(n):
        vals[i] = lib().oracle_rng_next(st.ctypes.data)
    return vals, st


def rng_init(seed, source, sample_no):
    st = np.zeros(4, "<u4")
    lib().oracle_rng_init(seed, source, sample_no, st.ctypes.data)
    return st


class Scene:
    def __init__(self, arrays):
        self.arrays = arrays  # keep alive
        d = ffi.desc_from_arrays(arrays)
        self._h = lib().oracle_scene_create(C.byref(d))
        if not self._h:
            raise ValueError("oracle_scene_create rejected the scene buffers")

    def close(self):
        if self._h:
            lib().oracle_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_closest(self, rays, want_counts=False):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        out = np.zeros(len(rays), ffi.HitId)
        c = ffi.TraversalCounts()
        lib().oracle_trace_closest(self._h, rays.ctypes.data, len(rays), out.ctypes.data, C.byref(c))
        return (out, c) if want_counts else out

    def trace_any(self, rays, tmax, want_counts=False):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        tmax = np.ascontiguousarray(tmax, "<f4")
        out = np.zeros(len(rays), ffi.HitId)
        c = ffi.TraversalCounts()
        lib().oracle_trace_any(self._h, rays.ctypes.data, tmax.ctypes.data, len(rays), out.ctypes.data, C.byref(c))
        return (out, c) if want_counts else out

    def trace_brute(self, rays):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        out = np.zeros(len(rays), ffi.HitId)
        lib().oracle_trace_brute(self._h, rays.ctypes.data, len(rays), out.ctypes.data)
        return out

    def render(self, camera, sources, width, height, params, first_sample, spp, threads=0, sum4=None, count=None):
        n = len(sources)
        sum4 = np.zeros(4 * n, "<u4") if sum4 is None else sum4
        count = np.zeros(n, "<u4") if count is None else count
        st = RenderStats()
        lib().oracle_render(self._h, camera.ctypes.data, sources.ctypes.data, n, width, height, C.byref(params),
                            first_sample, spp, sum4.ctypes.data, count.ctypes.data, C.byref(st), threads)
        return sum4, count, st

    def render_wavefront(self, camera, sources, width, height, params, first_sample, spp, slots):
        n = len(sources)
        sum4 = np.zeros(4 * n, "<u4")
        count = np.zeros(n, "<u4")
        st = RenderStats()
        frames = C.c_uint64()
        lib().oracle_render_wavefront(self._h, camera.ctypes.data, sources.ctypes.data, n, width, height,
                                      C.byref(params), first_sample, spp, slots, sum4.ctypes.data,
                                      count.ctypes.data, C.byref(st), C.byref(frames))
        return sum4, count, st, frames.value

    def render_active(self, camera, sources, width, height, params, first_sample, spp, active, sum4, count, moments,
                      threads=0):
        """spp samples for every source listed in `active` (None: all), added into sum4 / count / moments."""
        n = len(sources)
        act = None if active is None else np.ascontiguousarray(active, "<u4")
        st = RenderStats()
        lib().oracle_render_active(self._h, camera.ctypes.data, sources.ctypes.data, n, width, height, C.byref(params),
                                   first_sample, spp, None if act is None else act.ctypes.data,
                                   n if act is None else len(act), sum4.ctypes.data, count.ctypes.data,
                                   None if moments is None else moments.ctypes.data, C.byref(st), threads)
        return st

    def wavefront_per_slot(self, camera, sources, width, height, params, slot_rng, sample_limit, n_frames, slots):
        """The wavefront with one persistent random stream per slot (the reference's scheme)."""
        n = len(sources)
        rng = np.ascontiguousarray(slot_rng, "<u4").reshape(-1, 4)
        assert len(rng) == slots
        sum4 = np.zeros(4 * n, "<u4")
        count = np.zeros(n, "<u4")
        st = RenderStats()
        lib().oracle_wavefront_per_slot(self._h, camera.ctypes.data, sources.ctypes.data, n, width, height,
                                        C.byref(params), rng.ctypes.data, sample_limit, n_frames, slots,
                                        sum4.ctypes.data, count.ctypes.data, C.byref(st))
        return sum4, count, st

    def step_wavefront(self, camera, sources, width, height, params, first_sample, n_frames, slots):
        """Sums, counts and statistics after `n_frames` frames of the regenerating wavefront."""
        n = len(sources)
        sum4 = np.zeros(4 * n, "<u4")
        count = np.zeros(n, "<u4")
        st = RenderStats()
        lib().oracle_step_wavefront(self._h, camera.ctypes.data, sources.ctypes.data, n, width, height,
                                    C.byref(params), first_sample, n_frames, slots, sum4.ctypes.data,
                                    count.ctypes.data, C.byref(st))
        return sum4, count, st

    def bounce_rays(self, camera, sources, width, height, params, sample_no, bounce):
        out = np.zeros(len(sources), ffi.Ray)
        lib().oracle_bounce_rays(self._h, camera.ctypes.data, sources.ctypes.data, len(sources), width, height,
                                 C.byref(params), sample_no, bounce, out.ctypes.data)
        return out


def _connect_rays(self, camera, sources, width, height, params, sample_no, bounce):
    """Connect (shadow) rays traced at the hit of `bounce`: (rays, tmax) of the paths that trace one."""
    out = np.zeros(len(sources), ffi.Ray)
    tmax = np.zeros(len(sources), "<f4")
    lib().oracle_connect_rays(self._h, camera.ctypes.data, sources.ctypes.data, len(sources), width, height,
                              C.byref(params), sample_no, bounce, out.ctypes.data, tmax.ctypes.data)
    keep = out["last_inst"] != ffi.NO_HIT
    return out[keep], tmax[keep]


Scene.connect_rays = _connect_rays


def adapt_update(count, moments, flags, rel_threshold, min_samples):
    """Flags converged sources in place; returns the number still unflagged."""
    return lib().oracle_adapt_update(count.ctypes.data, moments.ctypes.data, flags.ctypes.data, len(count),
                                     rel_threshold, min_samples)


def resolve(sum4, count, sources, width, height, params):
    rgb = np.zeros(width * height, "<u4")
    lib().oracle_resolve(sum4.ctypes.data, count.ctypes.data, sources.ctypes.data, len(sources), width, height,
                         C.byref(params), rgb.ctypes.data)
    return rgb


def math_selftest(a, b, c):
    out = np.zeros(6 * len(a), "<f4")
    lib().oracle_math_selftest(a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a), out.ctypes.data)
    return out
