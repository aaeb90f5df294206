"""ctypes / numpy declarations of the C ABI in include/mirt.h and include/mirt_host.h.

Nothing here computes anything: it declares struct layouts (as numpy dtypes for
array arguments and ctypes Structures for by-pointer arguments) and loads the
two shared libraries.  The device library is required for every product path;
loading it fails loudly when it has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# ----------------------------------------------------------------- POD dtypes
Vertex = np.dtype([("position", "<f4", 4), ("normal", "<f4", 4), ("uv", "<f4", 4)])
Index = np.dtype([("v", "<u4", 4)])
BVHNode = np.dtype([("lb", "<f4", 4), ("ub", "<f4", 4), ("left", "<u4"), ("right", "<u4"),
                    ("is_leaf", "<u4"), ("start", "<u4"), ("end", "<u4"), ("pad", "<u4", 3)])
GeometryOffsets = np.dtype([("vertex", "<u4"), ("index", "<u4"), ("blas_node", "<u4")])
Instance = np.dtype([("transform", "<u4"), ("geometry", "<u4"), ("material", "<u4")])
Transform = np.dtype([("scale", "<f4", 4), ("rotation", "<f4", 4), ("translation", "<f4", 4)])
Material = np.dtype([("colour_texture", "<u4"), ("emissive_texture", "<u4"),
                     ("metallic_roughness_texture", "<u4"), ("normal_texture", "<u4"),
                     ("colour", "<f4", 4), ("emissive", "<f4", 4), ("metallic", "<f4"),
                     ("roughness", "<f4"), ("ior", "<f4"), ("transmission", "<f4")])
Camera = np.dtype([("position", "<f4", 3), ("pad0", "<u4"), ("forward", "<f4", 3), ("pad1", "<u4"),
                   ("up", "<f4", 3), ("pad2", "<u4"), ("dims", "<f4", 2), ("focal_length", "<f4"),
                   ("changed", "<u4")])
SampleSource = np.dtype([("screen_pos", "<f4", 2), ("out_pos", "<u4", 2), ("sample_count", "<u4"),
                         ("flags", "<u4")])
LambertianData = np.dtype([("albedo", "<f4", 4)])
MetallicData = np.dtype([("albedo", "<f4", 4), ("fuzz", "<f4"), ("pad", "<u4", 3)])
DielectricData = np.dtype([("albedo", "<f4", 4), ("ir", "<f4"), ("pad", "<u4", 3)])
EmissiveData = np.dtype([("albedo", "<f4", 4)])
ClassicMaterialRef = np.dtype([("kind", "<u4"), ("data", "<u4")])
Texture = np.dtype([("width", "<u4"), ("height", "<u4"), ("first_texel", "<u4"), ("filter", "<u4")])
GltfInfo = np.dtype([("n_meshes", "<u4"), ("n_instances", "<u4"), ("n_materials", "<u4"), ("n_textures", "<u4"),
                     ("n_baked", "<u4"), ("n_skipped_primitives", "<u4"), ("n_skipped_images", "<u4"), ("reserved", "<u4")])
FILTER_NEAREST, FILTER_BILINEAR = 0, 1
Ray = np.dtype([("pos", "<f4", 3), ("dir", "<f4", 3), ("last_inst", "<u4"), ("last_prim", "<u4")])
HitId = np.dtype([("instance_id", "<u4"), ("triangle_id", "<u4"), ("t", "<f4"), ("front_face", "<u4")])

assert Vertex.itemsize == 48 and Index.itemsize == 16 and BVHNode.itemsize == 64
assert GeometryOffsets.itemsize == 12 and Instance.itemsize == 12 and Transform.itemsize == 48
assert Material.itemsize == 64 and Camera.itemsize == 64 and SampleSource.itemsize == 24
assert Ray.itemsize == 32 and HitId.itemsize == 16

NO_HIT = 0xFFFFFFFF
NO_PRIM = 0x7FFFFFFF
SHADE_UBER, SHADE_CLASSIC = 0, 1
RNG_PER_SAMPLE, RNG_PER_SLOT = 0, 1
MAT_LAMBERTIAN, MAT_METALLIC, MAT_DIELECTRIC, MAT_EMISSIVE = 1, 2, 3, 4


# ------------------------------------------------------------ ctypes structs
class SceneDesc(C.Structure):
    _fields_ = [
        ("instances", C.c_void_p), ("n_instances", C.c_uint32),
        ("geometry_offsets", C.c_void_p), ("n_geometries", C.c_uint32),
        ("vertices", C.c_void_p), ("n_vertices", C.c_uint32),
        ("indices", C.c_void_p), ("n_indices", C.c_uint32),
        ("blas_nodes", C.c_void_p), ("n_blas_nodes", C.c_uint32),
        ("materials", C.c_void_p), ("n_materials", C.c_uint32),
        ("transforms", C.c_void_p), ("n_transforms", C.c_uint32),
        ("tlas_nodes", C.c_void_p), ("n_tlas_nodes", C.c_uint32),
        ("tlas_to_instances", C.c_void_p), ("n_tlas_to_instances", C.c_uint32),
        ("light_sources", C.c_void_p), ("n_light_sources", C.c_uint32),
        ("classic_refs", C.c_void_p), ("n_classic_refs", C.c_uint32),
        ("lambertian", C.c_void_p), ("n_lambertian", C.c_uint32),
        ("metallic", C.c_void_p), ("n_metallic", C.c_uint32),
        ("dielectric", C.c_void_p), ("n_dielectric", C.c_uint32),
        ("emissive", C.c_void_p), ("n_emissive", C.c_uint32),
        ("textures", C.c_void_p), ("n_textures", C.c_uint32),
        ("texels", C.c_void_p), ("n_texels", C.c_uint32),
    ]


# (field, count field, dtype) for every array of a SceneDesc
SCENE_ARRAYS = [
    ("instances", "n_instances", Instance),
    ("geometry_offsets", "n_geometries", GeometryOffsets),
    ("vertices", "n_vertices", Vertex),
    ("indices", "n_indices", Index),
    ("blas_nodes", "n_blas_nodes", BVHNode),
    ("materials", "n_materials", Material),
    ("transforms", "n_transforms", Transform),
    ("tlas_nodes", "n_tlas_nodes", BVHNode),
    ("tlas_to_instances", "n_tlas_to_instances", np.dtype("<u4")),
    ("light_sources", "n_light_sources", np.dtype("<u4")),
    ("classic_refs", "n_classic_refs", ClassicMaterialRef),
    ("lambertian", "n_lambertian", LambertianData),
    ("metallic", "n_metallic", MetallicData),
    ("dielectric", "n_dielectric", DielectricData),
    ("emissive", "n_emissive", EmissiveData),
    ("textures", "n_textures", Texture),
    ("texels", "n_texels", np.dtype("<u4")),
]


class Params(C.Structure):
    _fields_ = [("max_bounces", C.c_uint32), ("sky_radiance", C.c_float),
                ("fixed_point_scale", C.c_float), ("exposure", C.c_float),
                ("shade_mode", C.c_uint32), ("rng_mode", C.c_uint32), ("seed", C.c_uint32),
                ("nee", C.c_uint32), ("adaptive", C.c_uint32), ("stage_kernels", C.c_uint32)]


class TraversalCounts(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("tlas_nodes", C.c_uint64), ("instance_visits", C.c_uint64),
                ("blas_nodes", C.c_uint64), ("triangles", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def algorithmic_bytes(self, per_ray):
        """Bytes of the reference-layout traffic model: `per_ray` fixed bytes per ray,
        64 B per node visit, 76 B per instance visit, 160 B per triangle test."""
        return (per_ray * self.rays + 64 * (self.tlas_nodes + self.blas_nodes)
                + 76 * self.instance_visits + 160 * self.triangles)


class Stats(C.Structure):
    _fields_ = [("extend_rays", C.c_uint64), ("connect_rays", C.c_uint64),
                ("shade_events", C.c_uint64), ("spawned_samples", C.c_uint64),
                ("finished_samples", C.c_uint64), ("frames", C.c_uint64),
                ("extend_launches", C.c_uint64), ("extend_ms", C.c_double),
                ("finish_launches", C.c_uint64), ("finish_ms", C.c_double),
                ("connect_launches", C.c_uint64), ("connect_ms", C.c_double),
                ("shade_launches", C.c_uint64), ("shade_ms", C.c_double),
                ("sample_launches", C.c_uint64), ("sample_ms", C.c_double),
                ("extend_counts", TraversalCounts), ("connect_counts", TraversalCounts),
                ("extend_tri_wave_lds", C.c_uint64), ("extend_tri_wave_mem", C.c_uint64),
                ("connect_tri_wave_lds", C.c_uint64), ("connect_tri_wave_mem", C.c_uint64),
                ("overflowed_sums", C.c_uint64),
                ("engine_phase", C.c_uint64 * 16)]


class ConfigInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("max_bounces", C.c_uint32), ("shade_mode", C.c_uint32), ("nee", C.c_uint32),
                ("procedural_dragon", C.c_uint32), ("seed", C.c_uint32)]


def desc_from_arrays(arrays):
    """Build a SceneDesc whose pointers refer to the numpy arrays in `arrays`
    (a dict keyed like SCENE_ARRAYS).  The caller keeps `arrays` alive."""
    d = SceneDesc()
    for name, count, dt in SCENE_ARRAYS:
        a = arrays.get(name)
        if a is None or len(a) == 0:
            setattr(d, name, None)
            setattr(d, count, 0)
            continue
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
        setattr(d, name, a.ctypes.data)
        setattr(d, count, len(a))
    return d


def arrays_from_desc(d):
    """Copy every array a SceneDesc points at into owned numpy arrays."""
    out = {}
    for name, count, dt in SCENE_ARRAYS:
        n = getattr(d, count)
        ptr = getattr(d, name)
        if not ptr or n == 0:
            out[name] = np.zeros(0, dt)
            continue
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
        out[name] = np.frombuffer(buf, dtype=dt, count=n).copy()
    return out


# ------------------------------------------------------------------ loaders
_host = None
_device = None


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or `make` at the repository root)")
    return C.CDLL(path)


def host_lib():
    """libmirt_host.so — the host scene model (no GPU code)."""
    global _host
    if _host is None:
        L = _load("libmirt_host.so")
        vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
        L.mirth_scene_new.restype = vp
        L.mirth_scene_free.argtypes = [vp]
        L.mirth_last_error.restype = C.c_char_p
        L.mirth_last_error.argtypes = [vp]
        L.mirth_load_mesh_obj.argtypes = [vp, C.c_char_p]
        L.mirth_load_mesh_rect.argtypes = [vp]
        L.mirth_load_mesh_cube.argtypes = [vp]
        L.mirth_load_mesh_model.argtypes = [vp, C.c_char_p, vp, u32, vp, u32, vp, u32]
        L.mirth_load_mesh_model_file.argtypes = [vp, C.c_char_p]
        L.mirth_load_mesh_knot.argtypes = [vp, u32, u32, u32]
        L.mirth_load_mesh_subdivided.argtypes = [vp, C.c_char_p, C.c_char_p, u32, u32]
        L.mirth_set_dragon_standin.argtypes = [vp, i32, u32]
        L.mirth_add_texture.argtypes = [vp, u32, u32, vp, u32]
        L.mirth_mesh_set_planar_uv.argtypes = [vp, i32, i32, i32, C.c_float, C.c_float]
        L.mirth_add_texture_png.argtypes = [vp, vp, C.c_size_t, u32]
        L.mirth_load_gltf.argtypes = [vp, C.c_char_p, vp]
        L.mirth_det_sincos.restype = None
        L.mirth_det_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mirth_add_material.argtypes = [vp, vp]
        L.mirth_default_material.argtypes = [vp]
        L.mirth_default_camera.argtypes = [vp]
        L.mirth_spawn.argtypes = [vp, vp, i32, i32]
        L.mirth_spawn_cornell.argtypes = [vp, vp, vp, i32]
        L.mirth_build_config.argtypes = [vp, i32, C.c_char_p, u32, u32, C.POINTER(ConfigInfo)]
        L.mirth_bind.argtypes = [vp]
        L.mirth_scene_desc.argtypes = [vp, C.POINTER(SceneDesc)]
        L.mirth_make_sample_sources.argtypes = [u32, u32, u64, i32, vp]
        L.mirth_obj_to_model_file.argtypes = [C.c_char_p, C.c_char_p]
        L.mirth_parse_obj_counts.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u32), C.POINTER(u32),
                                             C.POINTER(u32)]
        L.mirth_set_blas_builder.argtypes = [vp, vp, vp]
        L.mirth_build_blas_host.argtypes = [vp, u32, vp, u32, vp, u32, C.POINTER(u32)]
        L.mirth_mesh_counts.argtypes = [vp, i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.mirth_mesh_copy.argtypes = [vp, i32, vp, vp, vp, vp]
        _host = L
    return _host


DEVICE_SYMBOLS = [
    "mirt_create", "mirt_destroy", "mirt_last_error", "mirt_status_string", "mirt_default_params",
    "mirt_scene_upload", "mirt_scene_update_instances", "mirt_scene_update_transforms", "mirt_state_create", "mirt_camera_set", "mirt_reset_accum",
    "mirt_set_sample_range", "mirt_state_set_rng", "mirt_adapt_update", "mirt_read_flags", "mirt_read_moments",
    "mirt_write_moments", "mirt_step", "mirt_render_spp", "mirt_synchronize", "mirt_read_output",
    "mirt_read_accum", "mirt_copy_accum_to_device", "mirt_load_accum_from_device",
    "mirt_trace_closest", "mirt_trace_closest_static", "mirt_trace_any", "mirt_bench_trace_closest", "mirt_bench_gather", "mirt_scene_bytes", "mirt_stats",
    "mirt_stats_reset", "mirt_set_counting", "mirt_set_event_timing", "mirt_dump_extension_rays",
    "mirt_math_selftest", "mirt_build_blas", "mirt_build_tlas",
    "mirt_group_create", "mirt_group_destroy", "mirt_group_size", "mirt_group_member", "mirt_group_last_error",
    "mirt_group_scene_upload", "mirt_group_scene_update_instances", "mirt_group_state_create",
    "mirt_group_camera_set", "mirt_group_reset_accum", "mirt_group_render_spp", "mirt_group_reduce",
    "mirt_group_adapt_update", "mirt_group_scene_update_transforms",
]


def device_lib():
    """libmirt.so — HIP kernels + C ABI.  Required: there is no CPU fallback."""
    global _device
    if _device is None:
        # MIRT_DEVICE_LIB selects another build of the same library (A/B experiments)
        L = _load(os.environ.get("MIRT_DEVICE_LIB", "libmirt.so"))
        vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
        for s in DEVICE_SYMBOLS:
            getattr(L, s)  # raises AttributeError if the ABI is incomplete
        L.mirt_create.argtypes = [i32, C.POINTER(vp)]
        L.mirt_destroy.argtypes = [vp]
        L.mirt_last_error.restype = C.c_char_p
        L.mirt_last_error.argtypes = [vp]
        L.mirt_status_string.restype = C.c_char_p
        L.mirt_status_string.argtypes = [i32]
        L.mirt_default_params.argtypes = [C.POINTER(Params)]
        L.mirt_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
        L.mirt_scene_update_instances.argtypes = [vp, C.POINTER(SceneDesc)]
        L.mirt_scene_update_transforms.argtypes = [vp, vp, u32, C.POINTER(C.c_float)]
        L.mirt_state_create.argtypes = [vp, u32, u32, u32, vp, u32, C.POINTER(Params)]
        L.mirt_camera_set.argtypes = [vp, vp]
        L.mirt_reset_accum.argtypes = [vp]
        L.mirt_set_sample_range.argtypes = [vp, u32]
        L.mirt_state_set_rng.argtypes = [vp, vp, u32]
        L.mirt_adapt_update.argtypes = [vp, C.c_float, u32, C.POINTER(u32)]
        L.mirt_read_flags.argtypes = [vp, vp, sz]
        L.mirt_read_moments.argtypes = [vp, vp, sz]
        L.mirt_write_moments.argtypes = [vp, vp, sz]
        L.mirt_step.argtypes = [vp, u32]
        L.mirt_render_spp.argtypes = [vp, u32]
        L.mirt_synchronize.argtypes = [vp]
        L.mirt_read_output.argtypes = [vp, vp, sz]
        L.mirt_read_accum.argtypes = [vp, vp, vp, sz]
        L.mirt_copy_accum_to_device.argtypes = [vp, vp, vp, sz]
        L.mirt_load_accum_from_device.argtypes = [vp, vp, vp, sz]
        L.mirt_trace_closest.argtypes = [vp, vp, sz, vp, C.POINTER(TraversalCounts)]
        L.mirt_trace_closest_static.argtypes = [vp, vp, sz, vp, C.POINTER(TraversalCounts)]
        L.mirt_trace_any.argtypes = [vp, vp, vp, sz, vp, C.POINTER(TraversalCounts)]
        L.mirt_bench_trace_closest.argtypes = [vp, vp, sz, u32, C.POINTER(C.c_double)]
        L.mirt_bench_gather.argtypes = [vp, sz, u32, u32, C.POINTER(C.c_double)]
        L.mirt_scene_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mirt_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mirt_stats_reset.argtypes = [vp]
        L.mirt_set_counting.argtypes = [vp, i32]
        L.mirt_set_event_timing.argtypes = [vp, i32]
        L.mirt_dump_extension_rays.argtypes = [vp, vp, sz, C.POINTER(sz)]
        L.mirt_math_selftest.argtypes = [vp, vp, vp, vp, sz, vp]
        L.mirt_build_blas.argtypes = [vp, vp, u32, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(C.c_float)]
        L.mirt_build_tlas.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32, C.POINTER(u32), vp, C.POINTER(C.c_float)]
        L.mirt_group_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
        L.mirt_group_destroy.argtypes = [vp]
        L.mirt_group_size.argtypes = [vp]
        L.mirt_group_member.restype = vp
        L.mirt_group_member.argtypes = [vp, i32]
        L.mirt_group_last_error.restype = C.c_char_p
        L.mirt_group_last_error.argtypes = [vp]
        L.mirt_group_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
        L.mirt_group_scene_update_instances.argtypes = [vp, C.POINTER(SceneDesc)]
        L.mirt_group_state_create.argtypes = [vp, u32, u32, u32, vp, u32, C.POINTER(Params)]
        L.mirt_group_camera_set.argtypes = [vp, vp]
        L.mirt_group_reset_accum.argtypes = [vp]
        L.mirt_group_render_spp.argtypes = [vp, u32]
        L.mirt_group_reduce.argtypes = [vp]
        L.mirt_group_adapt_update.argtypes = [vp, C.c_float, u32, C.POINTER(u32)]
        L.mirt_group_scene_update_transforms.argtypes = [vp, vp, u32, C.POINTER(C.c_float)]
        _device = L
    return _device
