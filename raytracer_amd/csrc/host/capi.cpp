// capi.cpp — extern "C" surface of the host scene model and the five
// benchmark scene configurations.
#include <cstdio>
#include <cstring>
#include <memory>
#include <sys/stat.h>

#include "host.hpp"
#include "mirt_host.h"

using namespace mirth;

struct mirth_scene {
  Scene scene;
  std::string error;
  bool bound = false;
  // what stands in for the absent dragon asset: 0 = the procedural knot,
  // 1 = suzanne with every triangle split into four `standin_levels` times
  int standin_kind = 0;
  uint32_t standin_levels = 5;
};

namespace {

// No exception crosses the C boundary: every entry point that returns a status is a function-try-block that
// ends here, with -1 and a message in the scene's error string when one can still be written.
int on_exception(mirth_scene *s) noexcept {
  try {
    std::string msg;
    try {
      throw;
    } catch (const std::bad_alloc &) {
      msg = "out of host memory";
    } catch (const std::exception &e) {
      msg = std::string("internal error: ") + e.what();
    } catch (...) {
      msg = "internal error";
    }
    if (s) s->error = msg;
  } catch (...) {
  }
  return -1;
}

const char kMagic[8] = {'M', 'I', 'R', 'T', 'M', 'E', 'S', 'H'};

// the id add_mesh returned, or -1 with the message of a failed builder callback
int added(mirth_scene *s, uint32_t id) {
  if (id != 0xFFFFFFFFu) return (int)id;
  s->error = s->scene.build_error;
  return -1;
}

bool write_model_file(const std::string &path, const ObjModel &m, std::string &err) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) {
    err = "cannot write " + path;
    return false;
  }
  uint32_t hdr[3] = {(uint32_t)m.positions.size(), (uint32_t)m.normals.size(), (uint32_t)m.indices.size()};
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, 4, 3, f) == 3 &&
            fwrite(m.positions.data(), 4, m.positions.size(), f) == m.positions.size() &&
            fwrite(m.normals.data(), 4, m.normals.size(), f) == m.normals.size() &&
            fwrite(m.indices.data(), 4, m.indices.size(), f) == m.indices.size();
  fclose(f);
  if (!ok) err = "short write to " + path;
  return ok;
}

bool read_model_file(const std::string &path, ObjModel &m, std::string &err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  char magic[8];
  uint32_t hdr[3];
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 && fread(hdr, 4, 3, f) == 3;
  if (ok) {
    m.positions.resize(hdr[0]);
    m.normals.resize(hdr[1]);
    m.indices.resize(hdr[2]);
    ok = fread(m.positions.data(), 4, hdr[0], f) == hdr[0] && fread(m.normals.data(), 4, hdr[1], f) == hdr[1] &&
         fread(m.indices.data(), 4, hdr[2], f) == hdr[2];
  }
  fclose(f);
  if (!ok) err = "bad MIRTMESH file " + path;
  if (ok) {
    for (uint32_t i : m.indices)
      if ((size_t)i * 3 + 2 >= m.positions.size()) {
        err = "index out of range in " + path;
        return false;
      }
  }
  return ok;
}

bool file_exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// Find <dir>/<name>.obj or <dir>/<name>.mirtmesh; returns a mesh id or -1.
int load_asset(mirth_scene *s, const std::string &dir, const std::string &name) {
  std::string obj = dir + "/" + name + ".obj";
  if (file_exists(obj)) return mirth_load_mesh_obj(s, obj.c_str());
  std::string mm = dir + "/" + name + ".mirtmesh";
  if (file_exists(mm)) return mirth_load_mesh_model_file(s, mm.c_str());
  s->error = "asset " + name + " not found in " + dir;
  return -1;
}

MirtTransform xform(float s, float rx, float ry, float rz, float tx, float ty, float tz) {
  MirtTransform t;
  t.scale[0] = t.scale[1] = t.scale[2] = s;
  t.scale[3] = s;
  t.rotation[0] = rx; t.rotation[1] = ry; t.rotation[2] = rz; t.rotation[3] = 0.0f;
  t.translation[0] = tx; t.translation[1] = ty; t.translation[2] = tz; t.translation[3] = 0.0f;
  return t;
}

MirtMaterial pbr(float r, float g, float b, float metallic, float roughness, float transmission, float ior) {
  MirtMaterial m = default_material();
  m.colour[0] = r; m.colour[1] = g; m.colour[2] = b; m.colour[3] = 1.0f;
  m.metallic = metallic;
  m.roughness = roughness;
  m.transmission = transmission;
  m.ior = ior;
  return m;
}

bool validate_model(const ObjModel &m, std::string &err) {
  if (m.positions.size() % 3 || m.normals.size() % 3 || m.indices.size() % 3) {
    err = "model arrays must have 3 entries per element";
    return false;
  }
  for (uint32_t i : m.indices)
    if ((size_t)i * 3 + 2 >= m.positions.size()) {
      err = "model index out of range";
      return false;
    }
  if (m.indices.empty()) {
    err = "model has no triangles";
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

mirth_scene *mirth_scene_new(void) { return new (std::nothrow) mirth_scene(); }
void mirth_scene_free(mirth_scene *s) { delete s; }
const char *mirth_last_error(const mirth_scene *s) { return s ? s->error.c_str() : "null scene"; }

int mirth_load_mesh_obj(mirth_scene *s, const char *path) try {
  if (!s || !path) return -1;
  std::string key = std::string("TOBJ:") + path;
  auto it = s->scene.by_desc.find(key);
  if (it != s->scene.by_desc.end()) return (int)it->second;
  ObjModel m;
  if (!load_obj_first_model(path, m, s->error)) return -1;
  return added(s, s->scene.add_mesh(key, Mesh::from_model(m)));
} catch (...) {
  return on_exception(s);
}

int mirth_load_mesh_rect(mirth_scene *s) try {
  return s ? added(s, s->scene.add_mesh("Rect", Mesh::rect())) : -1;
} catch (...) {
  return on_exception(s);
}
int mirth_load_mesh_cube(mirth_scene *s) try {
  return s ? added(s, s->scene.add_mesh("Cube", Mesh::cube())) : -1;
} catch (...) {
  return on_exception(s);
}

int mirth_load_mesh_model(mirth_scene *s, const char *name, const float *positions, uint32_t n_positions,
                          const float *normals, uint32_t n_normals, const uint32_t *indices,
                          uint32_t n_indices) try {
  if (!s || !name || !positions || !indices) return -1;
  ObjModel m;
  m.positions.assign(positions, positions + n_positions);
  if (normals) m.normals.assign(normals, normals + n_normals);
  m.indices.assign(indices, indices + n_indices);
  if (!validate_model(m, s->error)) return -1;
  return added(s, s->scene.add_mesh(std::string("MODEL:") + name, Mesh::from_model(m)));
} catch (...) {
  return on_exception(s);
}

int mirth_load_mesh_model_file(mirth_scene *s, const char *path) try {
  if (!s || !path) return -1;
  std::string key = std::string("MODELFILE:") + path;
  auto it = s->scene.by_desc.find(key);
  if (it != s->scene.by_desc.end()) return (int)it->second;
  ObjModel m;
  if (!read_model_file(path, m, s->error) || !validate_model(m, s->error)) return -1;
  return added(s, s->scene.add_mesh(key, Mesh::from_model(m)));
} catch (...) {
  return on_exception(s);
}

int mirth_load_mesh_knot(mirth_scene *s, uint32_t n_u, uint32_t n_v, uint32_t seed) try {
  if (!s || n_u < 3 || n_v < 3) return -1;
  std::string key = "KNOT:" + std::to_string(n_u) + "x" + std::to_string(n_v) + ":" + std::to_string(seed);
  auto it = s->scene.by_desc.find(key);
  if (it != s->scene.by_desc.end()) return (int)it->second;
  return added(s, s->scene.add_mesh(key, Mesh::from_model(make_knot_model(n_u, n_v, seed))));
} catch (...) {
  return on_exception(s);
}

int mirth_load_mesh_subdivided(mirth_scene *s, const char *asset_dir, const char *name, uint32_t levels, uint32_t seed) try {
  if (!s || !asset_dir || !name || levels > 8) return -1;
  const std::string dir = asset_dir;
  std::string key = std::string("SUBDIV:") + dir + "/" + name + ":" + std::to_string(levels) + ":" + std::to_string(seed);
  auto it = s->scene.by_desc.find(key);
  if (it != s->scene.by_desc.end()) return (int)it->second;
  ObjModel base;
  const std::string obj = dir + "/" + name + ".obj", mm = dir + "/" + name + ".mirtmesh";
  if (file_exists(obj)) {
    if (!load_obj_first_model(obj, base, s->error)) return -1;
  } else if (file_exists(mm)) {
    if (!read_model_file(mm, base, s->error)) return -1;
  } else {
    s->error = std::string("asset ") + name + " not found in " + dir;
    return -1;
  }
  if (!validate_model(base, s->error)) return -1;
  if ((base.indices.size() / 3) << (2 * levels) > (size_t)1 << 24) {
    s->error = "subdivided mesh would exceed 2^24 triangles";
    return -1;
  }
  return added(s, s->scene.add_mesh(key, Mesh::from_model(subdivide_model(base, levels, seed))));
} catch (...) {
  return on_exception(s);
}

void mirth_det_sincos(double x, double *s_out, double *c_out) {
  if (s_out) *s_out = host_det_sin(x);
  if (c_out) *c_out = host_det_cos(x);
}

int mirth_set_dragon_standin(mirth_scene *s, int kind, uint32_t levels) try {
  if (!s || kind < 0 || kind > 1 || levels > 8) return -1;
  s->standin_kind = kind;
  s->standin_levels = levels;
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_add_texture(mirth_scene *s, uint32_t width, uint32_t height, const uint32_t *rgba8, uint32_t filter) try {
  if (!s || !rgba8 || width == 0 || height == 0 || width > 65536u || height > 65536u || filter > MIRT_FILTER_BILINEAR)
    return -1;
  if (s->scene.texels.size() + (size_t)width * height > 0xFFFFFFFFull) return -1;
  MirtTexture t{width, height, (uint32_t)s->scene.texels.size(), filter};
  s->scene.texels.insert(s->scene.texels.end(), rgba8, rgba8 + (size_t)width * height);
  s->scene.textures.push_back(t);
  s->bound = false;
  return (int)s->scene.textures.size();  // 1-based: 0 means "no texture" in a material
} catch (...) {
  return on_exception(s);
}

int mirth_add_texture_png(mirth_scene *s, const void *png, size_t len, uint32_t filter) try {
  if (!s || !png || filter > MIRT_FILTER_BILINEAR) return -1;
  uint32_t w = 0, h = 0;
  std::vector<uint32_t> texels;
  if (!decode_png_rgba8(static_cast<const unsigned char *>(png), len, w, h, texels, s->error)) return -1;
  return mirth_add_texture(s, w, h, texels.data(), filter);
} catch (...) {
  return on_exception(s);
}

int mirth_load_gltf(mirth_scene *s, const char *path, MirthGltfInfo *info) try {
  if (!s || !path) return -1;
  MirthGltfInfo local;
  memset(&local, 0, sizeof(local));
  if (!load_gltf(s->scene, path, local, s->error)) return -1;
  s->bound = false;
  if (info) *info = local;
  return (int)local.n_instances;
} catch (...) {
  return on_exception(s);
}

int mirth_mesh_set_planar_uv(mirth_scene *s, int mesh_id, int axis_u, int axis_v, float scale, float offset) try {
  if (!s || mesh_id < 0 || (size_t)mesh_id >= s->scene.meshes.size() || axis_u < 0 || axis_u > 2 || axis_v < 0 || axis_v > 2)
    return -1;
  MeshData &d = s->scene.meshes[(size_t)mesh_id];
  d.uvs.resize(2 * d.mesh.positions.size());
  for (size_t i = 0; i < d.mesh.positions.size(); ++i) {
    const float *p = &d.mesh.positions[i].x;
    d.uvs[2 * i] = p[axis_u] * scale + offset;
    d.uvs[2 * i + 1] = p[axis_v] * scale + offset;
  }
  s->bound = false;
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_add_material(mirth_scene *s, const MirtMaterial *m) try {
  if (!s || !m) return -1;
  return (int)s->scene.add_material(*m);
} catch (...) {
  return on_exception(s);
}
void mirth_default_material(MirtMaterial *out) {
  if (out) *out = default_material();
}
void mirth_default_camera(MirtCamera *out) {
  if (out) *out = default_camera();
}

int mirth_spawn(mirth_scene *s, const MirtTransform *t, int mesh_id, int material_id) try {
  if (!s || !t || mesh_id < 0 || material_id < 0) return -1;
  if ((size_t)mesh_id >= s->scene.meshes.size() || (size_t)material_id >= s->scene.material_server.size()) {
    s->error = "spawn: unknown mesh or material id";
    return -1;
  }
  s->bound = false;
  return (int)s->scene.spawn(*t, (uint32_t)mesh_id, (uint32_t)material_id);
} catch (...) {
  return on_exception(s);
}

int mirth_spawn_cornell(mirth_scene *s, const float dims[3], const float pos[3], int with_ceiling_light) try {
  if (!s || !dims || !pos) return -1;
  s->bound = false;
  spawn_cornell(s->scene, {dims[0], dims[1], dims[2]}, {pos[0], pos[1], pos[2]}, with_ceiling_light != 0);
  if (!s->scene.build_error.empty()) {
    s->error = s->scene.build_error;
    return -1;
  }
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_build_blas_host(const float *positions4, uint32_t n_positions, uint32_t *faces4, uint32_t n_faces,
                          MirtBVHNode *nodes, uint32_t nodes_capacity, uint32_t *n_nodes) try {
  if (!n_nodes || !nodes || (n_faces != 0 && (!positions4 || !faces4))) return -1;
  for (size_t k = 0; k < (size_t)n_faces * 4; ++k)
    if (faces4[k] >= n_positions) return -1;
  Mesh m;
  m.positions.resize(n_positions);
  for (size_t i = 0; i < n_positions; ++i)
    m.positions[i] = {positions4[4 * i], positions4[4 * i + 1], positions4[4 * i + 2], positions4[4 * i + 3]};
  m.faces.assign(faces4, faces4 + (size_t)n_faces * 4);
  BLAS blas(std::move(m));
  *n_nodes = (uint32_t)blas.nodes.size();
  if (blas.nodes.size() > nodes_capacity) return -1;
  for (size_t i = 0; i < blas.nodes.size(); ++i) nodes[i] = to_gpu(blas.nodes[i]);
  if (n_faces) memcpy(faces4, blas.mesh.faces.data(), (size_t)n_faces * 16);
  return 0;
} catch (...) {
  return on_exception(nullptr);
}

int mirth_set_blas_builder(mirth_scene *s, mirth_blas_builder_fn fn, void *user) try {
  if (!s) return -1;
  s->scene.blas_builder = fn;
  s->scene.blas_builder_user = user;
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_build_config(mirth_scene *s, int config, const char *asset_dir, uint32_t dragon_n_u,
                       uint32_t dragon_n_v, MirthConfigInfo *info) try {
  if (!s || !info || config < 1 || config > 5) return -1;
  const std::string dir = asset_dir ? asset_dir : ".";
  if (dragon_n_u == 0) dragon_n_u = 2722;  // 2 * 2722 * 160 = 871 040 triangles
  if (dragon_n_v == 0) dragon_n_v = 160;
  memset(info, 0, sizeof(*info));
  info->seed = 0x5EED0000u + (uint32_t)config;
  info->shade_mode = MIRT_SHADE_UBER;
  Scene &sc = s->scene;
  s->bound = false;

  auto load_dragon = [&]() -> int {
    std::string obj = dir + "/dragon.obj", mm = dir + "/dragon.mirtmesh";
    if (file_exists(obj)) return mirth_load_mesh_obj(s, obj.c_str());
    if (file_exists(mm)) return mirth_load_mesh_model_file(s, mm.c_str());
    if (s->standin_kind == 1) {
      info->procedural_dragon = 2;
      return mirth_load_mesh_subdivided(s, dir.c_str(), "suzanne", s->standin_levels, 1);
    }
    info->procedural_dragon = 1;
    return mirth_load_mesh_knot(s, dragon_n_u, dragon_n_v, 1);
  };

  switch (config) {
    case 1: {
      // suzanne as one BLAS at (0,0,3) under a rect light
      int suz = load_asset(s, dir, "suzanne");
      if (suz < 0) return -1;
      int rect = mirth_load_mesh_rect(s);
      MirtMaterial grey = pbr(0.8f, 0.8f, 0.8f, 0.0f, 1.0f, 0.0f, 1.5f);  // src/scenes.rs:27-33
      MirtMaterial light = pbr(1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 0.0f, 1.5f); // src/scenes.rs:43-49
      light.colour[3] = 0.0f;
      light.emissive[0] = light.emissive[1] = light.emissive[2] = 20.0f;
      uint32_t g = sc.add_material(grey), l = sc.add_material(light);
      sc.spawn(xform(1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 3.0f), (uint32_t)suz, g);
      MirtTransform lt = xform(1.0f, 1.57079632679489661923f, 0.0f, 0.0f, 0.0f, 1.8f, 3.0f);
      lt.scale[0] = 2.0f;
      lt.scale[1] = 2.0f;
      sc.spawn(lt, (uint32_t)rect, l);
      info->width = 256; info->height = 256; info->spp = 1; info->max_bounces = 4;
      break;
    }
    case 2: {
      // open Cornell box + three teapots: Lambertian-like, Metallic, Dielectric
      int tea = load_asset(s, dir, "teapot");
      if (tea < 0) return -1;
      spawn_cornell(sc, {3.0f, 3.0f, 3.0f}, {0.0f, 0.0f, 3.0f}, false);
      uint32_t lam = sc.add_material(pbr(0.8f, 0.6f, 0.2f, 0.0f, 1.0f, 0.0f, 1.5f));
      uint32_t met = sc.add_material(pbr(0.9f, 0.9f, 0.9f, 1.0f, 0.01f, 0.0f, 1.5f));
      uint32_t die = sc.add_material(pbr(1.0f, 1.0f, 1.0f, 0.0f, 0.1f, 1.0f, 1.5f));
      sc.spawn(xform(0.45f, 0.0f, 0.6f, 0.0f, -0.95f, -1.05f, 3.3f), (uint32_t)tea, lam);
      sc.spawn(xform(0.45f, 0.0f, 0.0f, 0.0f, 0.0f, -1.05f, 2.6f), (uint32_t)tea, met);
      sc.spawn(xform(0.45f, 0.0f, -0.6f, 0.0f, 0.95f, -1.05f, 3.3f), (uint32_t)tea, die);
      info->width = 1920; info->height = 1080; info->spp = 64; info->max_bounces = 8;
      info->shade_mode = MIRT_SHADE_CLASSIC;
      break;
    }
    case 3:
    case 5: {
      // simple_scene (src/scenes.rs:136-193): cube mesh is loaded but unused,
      // three materials are registered before the Cornell box's five.
      mirth_load_mesh_cube(s);
      int dragon = load_dragon();
      if (dragon < 0) return -1;
      MirtMaterial gold = pbr(1.0f, 0.99f, 0.0f, 0.0f, 0.4f, 0.0f, 1.5f);
      MirtMaterial glass = pbr(1.0f, 1.0f, 1.0f, 0.2f, 0.1f, 0.0f, 1.5f);
      if (config == 5) {  // dielectric-heavy variant
        glass.metallic = 0.0f;
        glass.transmission = 1.0f;
      }
      MirtMaterial gray = pbr(0.8f, 0.8f, 0.8f, 0.0f, 1.0f, 0.0f, 1.5f);
      sc.add_material(gold);
      uint32_t glass_id = sc.add_material(glass);
      sc.add_material(gray);
      spawn_cornell(sc, {3.0f, 3.0f, 3.0f}, {0.0f, 0.0f, 3.0f}, config == 5);
      MirtTransform t = xform(1.0f, 0.0f, 0.0f, 0.0f, 0.0f, -0.89f, 2.75f);
      sc.spawn(t, (uint32_t)dragon, glass_id);
      if (config == 3) {
        info->width = 1920; info->height = 1080; info->spp = 256; info->max_bounces = 8;
      } else {
        info->width = 3840; info->height = 2160; info->spp = 1024; info->max_bounces = 8;
        info->shade_mode = MIRT_SHADE_CLASSIC;
        info->nee = 1;
      }
      break;
    }
    case 4: {
      // grid_scene (src/scenes.rs:676-786) widened to 10x10x10 instances,
      // meshes cycled dragon/teapot/suzanne/cube, seeded instead of OS entropy.
      int suz = load_asset(s, dir, "suzanne");
      int tea = load_asset(s, dir, "teapot");
      if (suz < 0 || tea < 0) return -1;
      int dragon = load_dragon();
      int cube = mirth_load_mesh_cube(s);
      if (dragon < 0) return -1;
      const int mesh_ids[4] = {dragon, tea, suz, cube};
      SplitMix64 rng(info->seed);
      std::vector<uint32_t> lam, met, die, emi;
      for (int i = 0; i < 15; ++i) lam.push_back(sc.add_material(pbr(rng.unit(), rng.unit(), rng.unit(), 0.0f, 1.0f, 0.0f, 1.5f)));
      for (int i = 0; i < 15; ++i) {
        float r = rng.unit(), g = rng.unit(), b = rng.unit();
        float fuzz = rng.range(-1.0f, 1.0f);
        fuzz = fuzz < 0.0f ? 0.0f : fuzz;
        met.push_back(sc.add_material(pbr(r, g, b, 1.0f, fuzz, 0.0f, 1.5f)));
      }
      for (int i = 0; i < 15; ++i) {
        MirtMaterial m = pbr(1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 0.0f, 1.5f);
        // the archive scene uses 800x emission against a dark sky; scaled to
        // the live sky radiance of 10 so lights and sky are comparable
        m.emissive[0] = rng.range(0.4f, 1.0f) * 20.0f;
        m.emissive[1] = rng.range(0.4f, 1.0f) * 20.0f;
        m.emissive[2] = rng.range(0.4f, 1.0f) * 20.0f;
        emi.push_back(sc.add_material(m));
      }
      for (int i = 0; i < 15; ++i) {
        float r = rng.unit(), g = rng.unit(), b = rng.unit();
        die.push_back(sc.add_material(pbr(r, g, b, 0.0f, 0.05f, 1.0f, rng.range(1.0f, 1.8f))));
      }
      int k = 0;
      for (int x = 1; x <= 10; ++x)
        for (int y = 0; y < 10; ++y)
          for (int z = 1; z <= 10; ++z) {
            float w = rng.range(0.0f, 4.1f);  // weights 2 : 1 : 1 : 0.1
            uint32_t mat = w < 2.0f ? lam[rng.below(15)]
                                    : (w < 3.0f ? met[rng.below(15)] : (w < 4.0f ? die[rng.below(15)] : emi[rng.below(15)]));
            int mesh = mesh_ids[(k++) & 3];
            float sc1 = rng.range(0.5f, 1.25f);
            float rx = rng.range(0.0f, 3.14159265f), ry = rng.range(0.0f, 3.14159265f), rz = rng.range(0.0f, 3.14159265f);
            float tx = (float)x * 2.0f + rng.range(-0.25f, 0.25f) - 11.0f;  // centred on the view axis
            float ty = (float)y * 2.0f + rng.range(-0.25f, 0.25f) - 9.0f;
            float tz = (float)z * 2.0f + rng.range(-0.25f, 0.25f) + 4.0f;
            sc.spawn(xform(sc1, rx, ry, rz, tx, ty, tz), (uint32_t)mesh, mat);
          }
      info->width = 1920; info->height = 1080; info->spp = 64; info->max_bounces = 8;
      break;
    }
  }
  if (!sc.build_error.empty()) {
    s->error = sc.build_error;
    return -1;
  }
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_bind(mirth_scene *s) try {
  if (!s) return -1;
  s->scene.bind();
  s->bound = true;
  if (s->scene.instances.empty()) {
    s->error = "bind: no instances";
    return -1;
  }
  return 0;
} catch (...) {
  return on_exception(s);
}

int mirth_scene_desc(const mirth_scene *s, MirtSceneDesc *out) try {
  if (!s || !out || !s->bound) return -1;
  s->scene.fill_desc(*out);
  return 0;
} catch (...) {
  return on_exception(const_cast<mirth_scene *>(s));
}

int mirth_make_sample_sources(uint32_t width, uint32_t height, uint64_t seed, int shuffle,
                              MirtSampleSource *out) try {
  if (!out || width == 0 || height == 0) return -1;
  std::vector<MirtSampleSource> v = make_sample_sources(width, height, seed, shuffle);
  memcpy(out, v.data(), v.size() * sizeof(MirtSampleSource));
  return 0;
} catch (...) {
  return on_exception(nullptr);
}

int mirth_obj_to_model_file(const char *obj_path, const char *out_path) try {
  if (!obj_path || !out_path) return -1;
  ObjModel m;
  std::string err;
  if (!load_obj_first_model(obj_path, m, err)) return -1;
  return write_model_file(out_path, m, err) ? 0 : -1;
} catch (...) {
  return on_exception(nullptr);
}

int mirth_parse_obj_counts(const char *text, size_t len, uint32_t *n_positions, uint32_t *n_normals,
                           uint32_t *n_triangles) try {
  if (!text) return -1;
  ObjModel m;
  std::string err;
  if (!parse_obj_first_model(text, len, m, err)) return -1;
  if (n_positions) *n_positions = (uint32_t)m.positions.size() / 3;
  if (n_normals) *n_normals = (uint32_t)m.normals.size() / 3;
  if (n_triangles) *n_triangles = (uint32_t)m.indices.size() / 3;
  return 0;
} catch (...) {
  return on_exception(nullptr);
}

int mirth_mesh_counts(const mirth_scene *s, int mesh_id, uint32_t *n_vertices, uint32_t *n_faces,
                      uint32_t *n_nodes) try {
  if (!s || mesh_id < 0 || (size_t)mesh_id >= s->scene.meshes.size()) return -1;
  const MeshData &d = s->scene.meshes[mesh_id];
  if (n_vertices) *n_vertices = (uint32_t)d.mesh.positions.size();
  if (n_faces) *n_faces = (uint32_t)d.mesh.n_faces();
  if (n_nodes) *n_nodes = (uint32_t)d.nodes.size();
  return 0;
} catch (...) {
  return on_exception(const_cast<mirth_scene *>(s));
}

int mirth_mesh_copy(const mirth_scene *s, int mesh_id, float *positions4, float *normals4, uint32_t *faces4,
                    MirtBVHNode *nodes) try {
  if (!s || mesh_id < 0 || (size_t)mesh_id >= s->scene.meshes.size()) return -1;
  const MeshData &d = s->scene.meshes[mesh_id];
  if (positions4) memcpy(positions4, d.mesh.positions.data(), d.mesh.positions.size() * 16);
  if (normals4) memcpy(normals4, d.mesh.normals.data(), d.mesh.normals.size() * 16);
  if (faces4) memcpy(faces4, d.mesh.faces.data(), d.mesh.faces.size() * 4);
  if (nodes) memcpy(nodes, d.nodes.data(), d.nodes.size() * sizeof(MirtBVHNode));
  return 0;
} catch (...) {
  return on_exception(const_cast<mirth_scene *>(s));
}

}  // extern "C"
