"""Device context: the role of the reference's PathtracerState + PathtracerPhase +
pathtracer_phase_execute (src/pathtracer_state.rs:96, src/pathtracer_manager.rs:74,195),
driven through the C ABI of libmirt.  Every call goes to the HIP library; there is
no CPU fallback — a missing library or device raises.
"""
import ctypes as C

import numpy as np

from . import ffi


class MirtError(RuntimeError):
    pass


def default_params(**kw):
    """MirtParams holding the reference's literals; keyword arguments override."""
    p = ffi.Params()
    ffi.device_lib().mirt_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Pathtracer:
    def __init__(self, device=0, _borrowed=None):
        self._L = ffi.device_lib()
        self._owned = _borrowed is None
        if _borrowed is not None:  # a member of a PathtracerGroup, which owns the context
            self._h = C.c_void_p(_borrowed)
        else:
            h = C.c_void_p()
            rc = self._L.mirt_create(device, C.byref(h))
            if rc != 0:
                raise MirtError(f"mirt_create(device={device}): {self._L.mirt_status_string(rc).decode()}")
            self._h = h
        self.n_sources = 0
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.mirt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise MirtError(f"{what}: {self._L.mirt_status_string(rc).decode()}: "
                            f"{self._L.mirt_last_error(self._h).decode()}")

    # ---- scene / state -------------------------------------------------
    def upload_scene(self, arrays):
        d = ffi.desc_from_arrays(arrays)
        self._check(self._L.mirt_scene_upload(self._h, C.byref(d)), "mirt_scene_upload")

    def update_instances(self, arrays):
        """Replace instances / transforms / materials / TLAS / lights; the geometry stays in HBM."""
        d = ffi.desc_from_arrays(arrays)
        self._check(self._L.mirt_scene_update_instances(self._h, C.byref(d)), "mirt_scene_update_instances")

    def update_transforms(self, transforms):
        """Only transforms moved: instance boxes, TLAS build and packing on the device.  Returns device ms."""
        t = np.ascontiguousarray(transforms, ffi.Transform)
        ms = C.c_float()
        self._check(self._L.mirt_scene_update_transforms(self._h, t.ctypes.data, len(t), C.byref(ms)),
                    "mirt_scene_update_transforms")
        return ms.value

    def create_state(self, width, height, threads, sources, params):
        sources = np.ascontiguousarray(sources, ffi.SampleSource)
        self._check(self._L.mirt_state_create(self._h, width, height, threads, sources.ctypes.data, len(sources),
                                              C.byref(params)), "mirt_state_create")
        self.n_sources = len(sources)
        self.width, self.height = width, height

    def set_camera(self, camera):
        cam = np.ascontiguousarray(camera, ffi.Camera)
        self._check(self._L.mirt_camera_set(self._h, cam.ctypes.data), "mirt_camera_set")

    def reset_accum(self):
        self._check(self._L.mirt_reset_accum(self._h), "mirt_reset_accum")

    def set_sample_range(self, first_sample):
        self._check(self._L.mirt_set_sample_range(self._h, first_sample), "mirt_set_sample_range")

    def set_rng(self, rng):
        """Per-slot random streams: one uint4 per path slot (the host's random_states vector)."""
        rng = np.ascontiguousarray(rng, "<u4").reshape(-1, 4)
        self._check(self._L.mirt_state_set_rng(self._h, rng.ctypes.data, len(rng)), "mirt_state_set_rng")

    # ---- adaptive sampling ---------------------------------------------
    def adapt_update(self, rel_threshold, min_samples):
        """Flag converged sources; returns how many sources still take samples."""
        n = C.c_uint32()
        self._check(self._L.mirt_adapt_update(self._h, rel_threshold, min_samples, C.byref(n)), "mirt_adapt_update")
        return n.value

    def read_flags(self):
        flags = np.zeros(self.n_sources, "<u4")
        self._check(self._L.mirt_read_flags(self._h, flags.ctypes.data, self.n_sources), "mirt_read_flags")
        return flags

    def read_moments(self):
        m = np.zeros(2 * self.n_sources, "<u8")
        self._check(self._L.mirt_read_moments(self._h, m.ctypes.data, self.n_sources), "mirt_read_moments")
        return m

    def write_moments(self, moments):
        m = np.ascontiguousarray(moments, "<u8")
        assert m.size == 2 * self.n_sources
        self._check(self._L.mirt_write_moments(self._h, m.ctypes.data, self.n_sources), "mirt_write_moments")

    # ---- rendering -----------------------------------------------------
    def step(self, n_frames=1):
        self._check(self._L.mirt_step(self._h, n_frames), "mirt_step")

    def render_spp(self, spp):
        self._check(self._L.mirt_render_spp(self._h, spp), "mirt_render_spp")

    def synchronize(self):
        self._check(self._L.mirt_synchronize(self._h), "mirt_synchronize")

    def read_output(self):
        rgb = np.zeros(self.width * self.height, "<u4")
        self._check(self._L.mirt_read_output(self._h, rgb.ctypes.data, rgb.size), "mirt_read_output")
        return rgb

    def read_accum(self):
        sum4 = np.zeros(4 * self.n_sources, "<u4")
        count = np.zeros(self.n_sources, "<u4")
        self._check(self._L.mirt_read_accum(self._h, sum4.ctypes.data, count.ctypes.data, self.n_sources),
                    "mirt_read_accum")
        return sum4, count

    def copy_accum_to_device(self, dev_sum4_ptr, dev_count_ptr):
        self._check(self._L.mirt_copy_accum_to_device(self._h, dev_sum4_ptr, dev_count_ptr, self.n_sources),
                    "mirt_copy_accum_to_device")

    def load_accum_from_device(self, dev_sum4_ptr, dev_count_ptr):
        self._check(self._L.mirt_load_accum_from_device(self._h, dev_sum4_ptr, dev_count_ptr, self.n_sources),
                    "mirt_load_accum_from_device")

    # ---- ray sets ------------------------------------------------------
    def trace_closest(self, rays, want_counts=False):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        out = np.zeros(len(rays), ffi.HitId)
        c = ffi.TraversalCounts()
        self._check(self._L.mirt_trace_closest(self._h, rays.ctypes.data, len(rays), out.ctypes.data,
                                               C.byref(c) if want_counts else None), "mirt_trace_closest")
        return (out, c) if want_counts else out

    def trace_closest_static(self, rays, want_counts=False):
        """mirt_trace_closest_static: the same query staged in the library's fixed device array (no allocation)."""
        rays = np.ascontiguousarray(rays, ffi.Ray)
        out = np.zeros(len(rays), ffi.HitId)
        c = ffi.TraversalCounts()
        self._check(self._L.mirt_trace_closest_static(self._h, rays.ctypes.data, len(rays), out.ctypes.data,
                                                      C.byref(c) if want_counts else None), "mirt_trace_closest_static")
        return (out, c) if want_counts else out

    def trace_any(self, rays, tmax, want_counts=False):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        tmax = np.ascontiguousarray(tmax, "<f4")
        out = np.zeros(len(rays), ffi.HitId)
        c = ffi.TraversalCounts()
        self._check(self._L.mirt_trace_any(self._h, rays.ctypes.data, tmax.ctypes.data, len(rays), out.ctypes.data,
                                           C.byref(c) if want_counts else None), "mirt_trace_any")
        return (out, c) if want_counts else out

    # ---- BLAS build ---------------------------------------------------
    def build_blas(self, positions4, faces4):
        """BLAS::new on the device: (reordered faces[m,4], nodes[k], device ms)."""
        pos = np.ascontiguousarray(positions4, "<f4").reshape(-1, 4)
        faces = np.array(faces4, "<u4").reshape(-1, 4)  # a copy: the build reorders it in place
        nodes = np.zeros(max(2 * len(faces), 1), ffi.BVHNode)
        n, ms = C.c_uint32(), C.c_float()
        self._check(self._L.mirt_build_blas(self._h, pos.ctypes.data, len(pos), faces.ctypes.data, len(faces),
                                            nodes.ctypes.data, len(nodes), C.byref(n), C.byref(ms)), "mirt_build_blas")
        return faces, nodes[:n.value].copy(), ms.value

    def build_tlas(self, geom_aabbs6, transforms, instances):
        """TLAS::new on the device: (nodes[k], instance_ids[n], device ms)."""
        a = np.ascontiguousarray(geom_aabbs6, "<f4").reshape(-1, 6)
        t = np.ascontiguousarray(transforms, ffi.Transform)
        inst = np.ascontiguousarray(instances, ffi.Instance)
        nodes = np.zeros(max(2 * len(inst), 1), ffi.BVHNode)
        ids = np.zeros(len(inst), "<u4")
        n, ms = C.c_uint32(), C.c_float()
        self._check(self._L.mirt_build_tlas(self._h, a.ctypes.data, len(a), t.ctypes.data, len(t), inst.ctypes.data,
                                            len(inst), nodes.ctypes.data, len(nodes), C.byref(n), ids.ctypes.data,
                                            C.byref(ms)), "mirt_build_tlas")
        return nodes[:n.value].copy(), ids, ms.value

    def blas_builder_callback(self):
        """(function pointer, user pointer) for Scene.set_blas_builder: mirt_build_blas on this context."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                            C.c_uint32, C.POINTER(C.c_uint32))
        L, h = self._L, self._h

        def on_device(user, pos, n_pos, faces, n_faces, nodes, cap, n_nodes):
            return L.mirt_build_blas(h, pos, n_pos, faces, n_faces, nodes, cap, n_nodes, None)

        return proto(on_device)

    def bench_trace_closest(self, rays, repeats=10):
        rays = np.ascontiguousarray(rays, ffi.Ray)
        ms = C.c_double()
        self._check(self._L.mirt_bench_trace_closest(self._h, rays.ctypes.data, len(rays), repeats, C.byref(ms)),
                    "mirt_bench_trace_closest")
        return ms.value

    def bench_gather(self, table_bytes=0, steps=512, repeats=3):
        """Lane-steps/s of dependent 32-byte gathers from a table (0 = the scene's size)."""
        r = C.c_double()
        self._check(self._L.mirt_bench_gather(self._h, table_bytes, steps, repeats, C.byref(r)), "mirt_bench_gather")
        return r.value

    def scene_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._L.mirt_scene_bytes(self._h, C.byref(a), C.byref(b)), "mirt_scene_bytes")
        return a.value, b.value

    def dump_extension_rays(self, capacity):
        out = np.zeros(capacity, ffi.Ray)
        n = C.c_size_t()
        self._check(self._L.mirt_dump_extension_rays(self._h, out.ctypes.data, capacity, C.byref(n)),
                    "mirt_dump_extension_rays")
        return out[: n.value]

    # ---- diagnostics ---------------------------------------------------
    def stats(self):
        s = ffi.Stats()
        self._check(self._L.mirt_stats(self._h, C.byref(s)), "mirt_stats")
        return s

    def stats_reset(self):
        self._check(self._L.mirt_stats_reset(self._h), "mirt_stats_reset")

    def set_counting(self, enabled):
        self._check(self._L.mirt_set_counting(self._h, int(enabled)), "mirt_set_counting")

    def set_event_timing(self, enabled):
        self._check(self._L.mirt_set_event_timing(self._h, int(enabled)), "mirt_set_event_timing")

    def math_selftest(self, a, b, c):
        a = np.ascontiguousarray(a, "<f4")
        b = np.ascontiguousarray(b, "<f4")
        c = np.ascontiguousarray(c, "<f4")
        out = np.zeros(6 * len(a), "<f4")
        self._check(self._L.mirt_math_selftest(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a),
                                               out.ctypes.data), "mirt_math_selftest")
        return out


def partition_samples(spp, world_size, rank):
    """Samples-per-pixel partition over GPUs: rank r renders sample numbers
    [first, first + n) of every source.  The union over ranks is [0, spp)."""
    base, extra = divmod(spp, world_size)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n


class PathtracerGroup:
    """Several GPUs driven from one process (mirt_group_*): scene, state and camera are
    replicated, a render call splits its samples-per-pixel over the members, reduce() adds
    the members' accumulators into member 0, which the reads go through."""

    def __init__(self, devices):
        self._L = ffi.device_lib()
        ords = (C.c_int32 * len(devices))(*devices)
        g = C.c_void_p()
        rc = self._L.mirt_group_create(ords, len(devices), C.byref(g))
        if rc != 0:
            raise MirtError(f"mirt_group_create({list(devices)}): {self._L.mirt_status_string(rc).decode()}")
        self._g = g
        self.members = [Pathtracer(_borrowed=self._L.mirt_group_member(g, i)) for i in range(len(devices))]

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()
            self._L.mirt_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __len__(self):
        return self._L.mirt_group_size(self._g)

    def _check(self, rc, what):
        if rc != 0:
            raise MirtError(f"{what}: {self._L.mirt_status_string(rc).decode()}: "
                            f"{self._L.mirt_group_last_error(self._g).decode()}")

    def upload_scene(self, arrays):
        d = ffi.desc_from_arrays(arrays)
        self._check(self._L.mirt_group_scene_upload(self._g, C.byref(d)), "mirt_group_scene_upload")

    def update_instances(self, arrays):
        d = ffi.desc_from_arrays(arrays)
        self._check(self._L.mirt_group_scene_update_instances(self._g, C.byref(d)), "mirt_group_scene_update_instances")

    def create_state(self, width, height, threads, sources, params):
        sources = np.ascontiguousarray(sources, ffi.SampleSource)
        self._check(self._L.mirt_group_state_create(self._g, width, height, threads, sources.ctypes.data, len(sources),
                                                    C.byref(params)), "mirt_group_state_create")
        for m in self.members:
            m.n_sources = len(sources)
            m.width, m.height = width, height

    def set_camera(self, camera):
        cam = np.ascontiguousarray(camera, ffi.Camera)
        self._check(self._L.mirt_group_camera_set(self._g, cam.ctypes.data), "mirt_group_camera_set")

    def reset_accum(self):
        self._check(self._L.mirt_group_reset_accum(self._g), "mirt_group_reset_accum")

    def render_spp(self, spp):
        self._check(self._L.mirt_group_render_spp(self._g, spp), "mirt_group_render_spp")

    def reduce(self):
        self._check(self._L.mirt_group_reduce(self._g), "mirt_group_reduce")

    def adapt_update(self, rel_threshold, min_samples):
        """Reduce, flag converged sources from the totals, same flags on every member; returns the sources still active."""
        n = C.c_uint32()
        self._check(self._L.mirt_group_adapt_update(self._g, rel_threshold, min_samples, C.byref(n)), "mirt_group_adapt_update")
        return n.value

    def update_transforms(self, transforms):
        t = np.ascontiguousarray(transforms, ffi.Transform)
        ms = C.c_float()
        self._check(self._L.mirt_group_scene_update_transforms(self._g, t.ctypes.data, len(t), C.byref(ms)),
                    "mirt_group_scene_update_transforms")
        return ms.value

    def read_accum(self):
        return self.members[0].read_accum()

    def read_output(self):
        return self.members[0].read_output()
