"""Python face of the host scene model (libmirt_host).

Mirrors the reference's host API names — MeshServer.load_mesh
(src/mesh.rs:149), MaterialServer.add_material (src/material.rs:54),
commands.spawn((Transform, MaterialId, MeshId)), spawn_cornell
(src/scenes.rs:20), binder_system (src/binder.rs:51) — and hands back the flat
buffers of bind group 0 as numpy arrays.  All arithmetic happens in the C++
library; this file only moves pointers.
"""
import ctypes as C
import os

import numpy as np

from . import ffi

ASSET_DIRS = [
    os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"),
]


def default_asset_dir():
    """Directory holding suzanne/teapot (and optionally dragon) as .obj or .mirtmesh."""
    env = os.environ.get("MIRT_ASSET_DIR")
    if env:
        return env
    return ASSET_DIRS[0]


def default_material():
    m = np.zeros(1, ffi.Material)
    ffi.host_lib().mirth_default_material(m.ctypes.data)
    return m


def default_camera(width=None, height=None):
    """CameraData::new() (src/camera.rs:117-128); dims = (W/H, 1) when a size is given."""
    c = np.zeros(1, ffi.Camera)
    ffi.host_lib().mirth_default_camera(c.ctypes.data)
    if width is not None:
        c["dims"][0] = (np.float32(width) / np.float32(height), np.float32(1.0))
    return c


def make_transform(scale=(1, 1, 1), rotation=(0, 0, 0), translation=(0, 0, 0)):
    t = np.zeros(1, ffi.Transform)
    t["scale"][0, :3] = scale
    t["rotation"][0, :3] = rotation
    t["translation"][0, :3] = translation
    return t


def make_material(colour=(0.8, 0.8, 0.8, 1.0), emissive=(0, 0, 0, 0), metallic=0.0, roughness=1.0,
                  ior=1.5, transmission=0.0, colour_texture=0, emissive_texture=0):
    m = default_material()
    m["colour_texture"] = colour_texture
    m["emissive_texture"] = emissive_texture
    m["colour"][0] = colour
    m["emissive"][0] = emissive
    m["metallic"] = metallic
    m["roughness"] = roughness
    m["ior"] = ior
    m["transmission"] = transmission
    return m


def make_sample_sources(width, height, seed, shuffle=2):
    out = np.zeros(width * height, ffi.SampleSource)
    rc = ffi.host_lib().mirth_make_sample_sources(width, height, seed, shuffle, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("mirth_make_sample_sources failed")
    return out


def build_blas(positions4, faces4):
    """BLAS::new on the host, on positions as given: (reordered faces[m,4], nodes[k])."""
    L = ffi.host_lib()
    pos = np.ascontiguousarray(positions4, "<f4").reshape(-1, 4)
    faces = np.array(faces4, "<u4").reshape(-1, 4)  # a copy: the build reorders it in place
    nodes = np.zeros(max(2 * len(faces), 1), ffi.BVHNode)
    n = C.c_uint32()
    rc = L.mirth_build_blas_host(pos.ctypes.data, len(pos), faces.ctypes.data, len(faces), nodes.ctypes.data,
                                 len(nodes), C.byref(n))
    if rc != 0:
        raise RuntimeError("mirth_build_blas_host failed")
    return faces, nodes[:n.value].copy()


class HostError(RuntimeError):
    """A call into libmirt_host.so failed; the message is mirth_last_error's."""


class Scene:
    """MeshServer + MaterialServer + entity list + binder."""

    def __init__(self):
        self._L = ffi.host_lib()
        self._h = self._L.mirth_scene_new()
        if not self._h:
            raise MemoryError("mirth_scene_new")
        self.info = None

    def close(self):
        if self._h:
            self._L.mirth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise HostError(f"{what}: {self._L.mirth_last_error(self._h).decode()}")
        return rc

    def set_blas_builder(self, pathtracer):
        """Meshes loaded from now on get their BLAS from `pathtracer`'s device (None: the host builder)."""
        self._builder = None if pathtracer is None else pathtracer.blas_builder_callback()  # keeps the thunk alive
        fn = None if self._builder is None else C.cast(self._builder, C.c_void_p)
        self._check(self._L.mirth_set_blas_builder(self._h, fn, None), "set_blas_builder")

    # MeshServer ---------------------------------------------------------
    def load_obj(self, path):
        return self._check(self._L.mirth_load_mesh_obj(self._h, path.encode()), "load_obj")

    def load_rect(self):
        return self._check(self._L.mirth_load_mesh_rect(self._h), "load_rect")

    def load_cube(self):
        return self._check(self._L.mirth_load_mesh_cube(self._h), "load_cube")

    def load_model(self, name, positions, indices, normals=None):
        p = np.ascontiguousarray(positions, "<f4").reshape(-1)
        i = np.ascontiguousarray(indices, "<u4").reshape(-1)
        n = None if normals is None else np.ascontiguousarray(normals, "<f4").reshape(-1)
        return self._check(self._L.mirth_load_mesh_model(
            self._h, name.encode(), p.ctypes.data, p.size, None if n is None else n.ctypes.data,
            0 if n is None else n.size, i.ctypes.data, i.size), "load_model")

    def load_model_file(self, path):
        return self._check(self._L.mirth_load_mesh_model_file(self._h, path.encode()), "load_model_file")

    def load_knot(self, n_u, n_v, seed=1):
        return self._check(self._L.mirth_load_mesh_knot(self._h, n_u, n_v, seed), "load_knot")

    def load_subdivided(self, name, levels, seed=1, asset_dir=None):
        """An asset with every triangle split into four `levels` times (jittered midpoints)."""
        return self._check(self._L.mirth_load_mesh_subdivided(self._h, (asset_dir or default_asset_dir()).encode(),
                                                              name.encode(), levels, seed), "load_subdivided")

    def set_dragon_standin(self, kind, levels=5):
        """What replaces the absent dragon asset in build_config: 'knot' or 'suzanne' (subdivided `levels` times)."""
        self._check(self._L.mirth_set_dragon_standin(self._h, {"knot": 0, "suzanne": 1}[kind], levels),
                    "set_dragon_standin")

    def mesh(self, mesh_id):
        """(positions[n,4], normals[n,4], faces[m,4], nodes[k]) of a built mesh."""
        nv, nf, nn = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._L.mirth_mesh_counts(self._h, mesh_id, nv, nf, nn), "mesh_counts")
        pos = np.zeros((nv.value, 4), "<f4")
        nrm = np.zeros((nv.value, 4), "<f4")
        faces = np.zeros((nf.value, 4), "<u4")
        nodes = np.zeros(nn.value, ffi.BVHNode)
        self._check(self._L.mirth_mesh_copy(self._h, mesh_id, pos.ctypes.data, nrm.ctypes.data,
                                            faces.ctypes.data, nodes.ctypes.data), "mesh_copy")
        return pos, nrm, faces, nodes

    # textures -----------------------------------------------------------
    def add_texture(self, rgba8, filter=ffi.FILTER_BILINEAR):
        """rgba8[h, w] packed texels (R | G<<8 | B<<16 | A<<24); returns the 1-based texture id."""
        t = np.ascontiguousarray(rgba8, "<u4")
        h, w = t.shape
        return self._check(self._L.mirth_add_texture(self._h, w, h, t.ctypes.data, filter), "add_texture")

    def add_texture_png(self, png_bytes, filter=ffi.FILTER_BILINEAR):
        """A PNG image in memory; returns the 1-based texture id."""
        buf = C.create_string_buffer(bytes(png_bytes), len(png_bytes))
        return self._check(self._L.mirth_add_texture_png(self._h, buf, len(png_bytes), filter), "add_texture_png")

    def load_gltf(self, path):
        """Every triangle primitive of a glTF 2.0 scene (.gltf / .glb) as meshes, entities, materials and
        textures; returns what was done as a dict (n_meshes, n_instances, ...)."""
        info = np.zeros(1, ffi.GltfInfo)
        self._check(self._L.mirth_load_gltf(self._h, path.encode(), info.ctypes.data), "load_gltf")
        return {k: int(info[0][k]) for k in info.dtype.names if k != "reserved"}

    def set_planar_uv(self, mesh_id, axis_u=0, axis_v=1, scale=0.5, offset=0.5):
        """uv = position[axis] * scale + offset for every vertex of a loaded mesh."""
        self._check(self._L.mirth_mesh_set_planar_uv(self._h, mesh_id, axis_u, axis_v, scale, offset), "set_planar_uv")

    # MaterialServer -----------------------------------------------------
    def add_material(self, material):
        m = np.ascontiguousarray(material, ffi.Material)
        return self._check(self._L.mirth_add_material(self._h, m.ctypes.data), "add_material")

    # entities -----------------------------------------------------------
    def spawn(self, transform, material_id, mesh_id):
        t = np.ascontiguousarray(transform, ffi.Transform)
        return self._check(self._L.mirth_spawn(self._h, t.ctypes.data, mesh_id, material_id), "spawn")

    def spawn_cornell(self, dims=(3.0, 3.0, 3.0), pos=(0.0, 0.0, 3.0), ceiling_light=False):
        d = np.asarray(dims, "<f4")
        p = np.asarray(pos, "<f4")
        self._check(self._L.mirth_spawn_cornell(self._h, d.ctypes.data, p.ctypes.data, int(ceiling_light)),
                    "spawn_cornell")

    def build_config(self, config, asset_dir=None, dragon_n_u=0, dragon_n_v=0):
        info = ffi.ConfigInfo()
        self._check(self._L.mirth_build_config(self._h, config, (asset_dir or default_asset_dir()).encode(),
                                               dragon_n_u, dragon_n_v, C.byref(info)), "build_config")
        self.info = info
        return info

    # binder -------------------------------------------------------------
    def bind(self):
        """Pack the scene; returns a dict of owned numpy arrays keyed by buffer name."""
        self._check(self._L.mirth_bind(self._h), "bind")
        d = ffi.SceneDesc()
        self._check(self._L.mirth_scene_desc(self._h, C.byref(d)), "scene_desc")
        return ffi.arrays_from_desc(d)


def build_config_scene(config, asset_dir=None, dragon_n_u=0, dragon_n_v=0, standin="knot", standin_levels=5):
    """(arrays, info) for benchmark configuration 1..5."""
    s = Scene()
    s.set_dragon_standin(standin, standin_levels)
    info = s.build_config(config, asset_dir, dragon_n_u, dragon_n_v)
    arrays = s.bind()
    s.close()
    return arrays, info
