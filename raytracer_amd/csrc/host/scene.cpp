// scene.cpp — MeshServer / MaterialServer / binder packing and the scene
// builders (src/mesh.rs:148-285, src/material.rs:47-71, src/binder.rs:184-282,
// src/scenes.rs:20-193, src/camera.rs:117-128, src/pathtracer_state.rs:175-195).
#include <algorithm>
#include <cmath>
#include <cstring>

#include <string>

#include <unordered_map>

#include "host.hpp"

namespace mirth {

MirtMaterial default_material() {  // src/material.rs:27-42
  MirtMaterial m;
  memset(&m, 0, sizeof(m));
  m.ior = 1.5f;
  return m;
}

MirtCamera default_camera() {  // src/camera.rs:117-128
  MirtCamera c;
  memset(&c, 0, sizeof(c));
  c.forward[2] = 1.0f;
  c.up[1] = 1.0f;
  c.dims[0] = 1.0f;
  c.dims[1] = 1.0f;
  c.focal_length = 1.0f;
  return c;
}

uint32_t Scene::add_mesh(const std::string &desc, Mesh mesh) {
  // MeshServer::load_mesh de-duplicates by descriptor (src/mesh.rs:149-165);
  // the worker then builds the BLAS (src/mesh.rs:118-145).
  auto it = by_desc.find(desc);
  if (it != by_desc.end()) return it->second;
  MeshData d;
  if (blas_builder != nullptr && mesh.n_faces() != 0) {
    // the caller's builder (the device one) stands in for BLAS::new
    const uint32_t nf = (uint32_t)mesh.n_faces();
    d.nodes.resize(2 * (size_t)nf);
    uint32_t n_nodes = 0;
    const int rc = blas_builder(blas_builder_user, &mesh.positions[0].x, (uint32_t)mesh.positions.size(),
                                mesh.faces.data(), nf, d.nodes.data(), (uint32_t)d.nodes.size(), &n_nodes);
    if (rc != 0 || n_nodes == 0 || n_nodes > d.nodes.size()) {
      build_error = "add_mesh: the BLAS builder callback failed with status " + std::to_string(rc);
      return 0xFFFFFFFFu;
    }
    d.nodes.resize(n_nodes);
    d.aabb = AABB{{d.nodes[0].lb[0], d.nodes[0].lb[1], d.nodes[0].lb[2]}, {d.nodes[0].ub[0], d.nodes[0].ub[1], d.nodes[0].ub[2]}};
    d.mesh = std::move(mesh);
  } else {
    BLAS blas(std::move(mesh));
    d.aabb = blas.nodes[0].bounds;
    d.nodes.reserve(blas.nodes.size());
    for (const BVHNode &n : blas.nodes) d.nodes.push_back(to_gpu(n));
    d.mesh = std::move(blas.mesh);
  }
  meshes.push_back(std::move(d));
  uint32_t id = (uint32_t)meshes.size() - 1;
  by_desc.emplace(desc, id);
  return id;
}

uint32_t Scene::add_material(const MirtMaterial &m) {  // src/material.rs:54-57
  material_server.push_back(m);
  return (uint32_t)material_server.size() - 1;
}

uint32_t Scene::spawn(const MirtTransform &t, uint32_t mesh_id, uint32_t material_id) {
  entities.push_back({t, mesh_id, material_id});
  return (uint32_t)entities.size() - 1;
}

void Scene::regenerate_buffer() {  // src/mesh.rs:198-284
  vertices.clear();
  indices.clear();
  blas_nodes.clear();
  geometry_offsets.clear();
  aabbs.clear();
  for (const MeshData &md : meshes) {
    geometry_offsets.push_back({(uint32_t)vertices.size(), (uint32_t)indices.size(), (uint32_t)blas_nodes.size()});
    aabbs.push_back(md.aabb);
    blas_nodes.insert(blas_nodes.end(), md.nodes.begin(), md.nodes.end());
    const size_t nv = std::min(md.mesh.positions.size(), md.mesh.normals.size());  // zip
    for (size_t i = 0; i < nv; ++i) {
      MirtVertex v;
      memset(&v, 0, sizeof(v));
      memcpy(v.position, &md.mesh.positions[i], 16);
      memcpy(v.normal, &md.mesh.normals[i], 16);
      if (2 * i + 1 < md.uvs.size()) {
        v.uv[0] = md.uvs[2 * i];
        v.uv[1] = md.uvs[2 * i + 1];
      }
      vertices.push_back(v);
    }
    for (size_t f = 0; f < md.mesh.n_faces(); ++f) {
      MirtIndex ix;
      memcpy(ix.v, &md.mesh.faces[4 * f], 16);
      indices.push_back(ix);
    }
  }
}

void Scene::bind() {  // src/binder.rs:184-282
  regenerate_buffer();
  instances.clear();
  materials.clear();
  transforms.clear();
  light_sources.clear();
  std::unordered_map<uint32_t, uint32_t> materials_id_map;
  std::vector<uint8_t> material_is_emissive;

  for (const Entity &e : entities) {
    if (e.mesh_id >= meshes.size()) continue;  // geom_id() == None
    uint32_t material_idx;
    auto it = materials_id_map.find(e.material_id);
    if (it != materials_id_map.end()) {
      material_idx = it->second;
    } else {
      if (e.material_id >= material_server.size()) continue;
      const MirtMaterial &m = material_server[e.material_id];
      bool emissive = m.emissive[0] != 0.0f || m.emissive[1] != 0.0f || m.emissive[2] != 0.0f ||
                      m.emissive[3] != 0.0f || m.emissive_texture > 0;
      materials.push_back(m);
      material_is_emissive.push_back(emissive ? 1 : 0);
      material_idx = (uint32_t)materials.size() - 1;
      materials_id_map.emplace(e.material_id, material_idx);
    }
    transforms.push_back(e.transform);
    MirtInstance inst{(uint32_t)transforms.size() - 1, e.mesh_id, material_idx};
    instances.push_back(inst);
    // The reference lists an instance as a light only the first time its
    // material is seen (src/binder.rs:213-243); every emissive instance is
    // listed here so that light sampling can reach all of them.
    if (material_is_emissive[material_idx]) light_sources.push_back((uint32_t)instances.size() - 1);
  }
  if (light_sources.empty()) light_sources.push_back(0xFFFFFFFFu);  // src/binder.rs:251-255

  tlas_nodes.clear();
  tlas_to_instances.clear();
  if (!instances.empty()) {
    TLAS tlas(aabbs, transforms, instances);
    for (size_t id : tlas.instance_ids) tlas_to_instances.push_back((uint32_t)id);
    for (const BVHNode &n : tlas.nodes) tlas_nodes.push_back(to_gpu(n));
  }

  // Tables for the per-material kernels.  The live Material has no kind tag,
  // so one is derived: emissive != 0 -> Emissive{emissive}; transmission > 0.5
  // -> Dielectric{colour, ior}; metallic > 0.5 -> Metallic{colour, fuzz =
  // roughness}; otherwise Lambertian{colour}.
  classic_refs.clear();
  lambertian.clear();
  metallic.clear();
  dielectric.clear();
  emissive.clear();
  for (size_t i = 0; i < materials.size(); ++i) {
    const MirtMaterial &m = materials[i];
    MirtClassicMaterialRef ref;
    if (material_is_emissive[i]) {
      MirtEmissiveData d;
      memcpy(d.albedo, m.emissive, 16);
      emissive.push_back(d);
      ref = {MIRT_MAT_EMISSIVE, (uint32_t)emissive.size() - 1};
    } else if (m.transmission > 0.5f) {
      MirtDielectricData d;
      memset(&d, 0, sizeof(d));
      memcpy(d.albedo, m.colour, 16);
      d.ir = m.ior;
      dielectric.push_back(d);
      ref = {MIRT_MAT_DIELECTRIC, (uint32_t)dielectric.size() - 1};
    } else if (m.metallic > 0.5f) {
      MirtMetallicData d;
      memset(&d, 0, sizeof(d));
      memcpy(d.albedo, m.colour, 16);
      d.fuzz = m.roughness;
      metallic.push_back(d);
      ref = {MIRT_MAT_METALLIC, (uint32_t)metallic.size() - 1};
    } else {
      MirtLambertianData d;
      memcpy(d.albedo, m.colour, 16);
      lambertian.push_back(d);
      ref = {MIRT_MAT_LAMBERTIAN, (uint32_t)lambertian.size() - 1};
    }
    classic_refs.push_back(ref);
  }
}

void Scene::fill_desc(MirtSceneDesc &d) const {
  memset(&d, 0, sizeof(d));
  d.instances = instances.data();                 d.n_instances = (uint32_t)instances.size();
  d.geometry_offsets = geometry_offsets.data();   d.n_geometries = (uint32_t)geometry_offsets.size();
  d.vertices = vertices.data();                   d.n_vertices = (uint32_t)vertices.size();
  d.indices = indices.data();                     d.n_indices = (uint32_t)indices.size();
  d.blas_nodes = blas_nodes.data();               d.n_blas_nodes = (uint32_t)blas_nodes.size();
  d.materials = materials.data();                 d.n_materials = (uint32_t)materials.size();
  d.transforms = transforms.data();               d.n_transforms = (uint32_t)transforms.size();
  d.tlas_nodes = tlas_nodes.data();               d.n_tlas_nodes = (uint32_t)tlas_nodes.size();
  d.tlas_to_instances = tlas_to_instances.data(); d.n_tlas_to_instances = (uint32_t)tlas_to_instances.size();
  d.light_sources = light_sources.data();         d.n_light_sources = (uint32_t)light_sources.size();
  d.classic_refs = classic_refs.data();           d.n_classic_refs = (uint32_t)classic_refs.size();
  d.lambertian = lambertian.data();               d.n_lambertian = (uint32_t)lambertian.size();
  d.metallic = metallic.data();                   d.n_metallic = (uint32_t)metallic.size();
  d.dielectric = dielectric.data();               d.n_dielectric = (uint32_t)dielectric.size();
  d.emissive = emissive.data();                   d.n_emissive = (uint32_t)emissive.size();
  d.textures = textures.data();                   d.n_textures = (uint32_t)textures.size();
  d.texels = texels.data();                       d.n_texels = (uint32_t)texels.size();
}

// ------------------------------------------------------------------- scenes

namespace {
MirtTransform make_transform(float sx, float sy, float sz, float sw, float rx, float ry, float rz,
                             float tx, float ty, float tz, float tw) {
  MirtTransform t;
  t.scale[0] = sx; t.scale[1] = sy; t.scale[2] = sz; t.scale[3] = sw;
  t.rotation[0] = rx; t.rotation[1] = ry; t.rotation[2] = rz; t.rotation[3] = 0.0f;
  t.translation[0] = tx; t.translation[1] = ty; t.translation[2] = tz; t.translation[3] = tw;
  return t;
}
MirtMaterial make_material(float r, float g, float b, float a, float metallic, float roughness) {
  MirtMaterial m = default_material();
  m.colour[0] = r; m.colour[1] = g; m.colour[2] = b; m.colour[3] = a;
  m.metallic = metallic;
  m.roughness = roughness;
  return m;
}
const float kPi = 3.14159265358979323846f;       // f32::consts::PI
const float kFracPi2 = 1.57079632679489661923f;  // f32::consts::FRAC_PI_2
}  // namespace

void spawn_cornell(Scene &s, Vec3 dims, Vec3 pos, bool with_ceiling_light) {
  // src/scenes.rs:20-134; the five materials are added in the same order.
  uint32_t gray = s.add_material(make_material(0.8f, 0.8f, 0.8f, 1.0f, 0.0f, 1.0f));
  uint32_t mirror = s.add_material(make_material(1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 0.01f));
  (void)mirror;
  MirtMaterial lm = make_material(1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 1.0f);
  lm.emissive[0] = lm.emissive[1] = lm.emissive[2] = 20.0f;
  lm.emissive[3] = 0.0f;
  uint32_t light = s.add_material(lm);
  uint32_t green = s.add_material(make_material(0.4f, 0.8f, 0.4f, 1.0f, 0.0f, 1.0f));
  uint32_t red = s.add_material(make_material(0.8f, 0.4f, 0.4f, 1.0f, 0.0f, 1.0f));
  uint32_t rect = s.add_mesh("Rect", Mesh::rect());

  // Floor (src/scenes.rs:70-78)
  s.spawn(make_transform(dims.x, dims.z, 1.0f, 0.0f, kFracPi2, 0.0f, 0.0f,
                         0.0f + pos.x, -dims.y / 2.0f + pos.y, 0.0f + pos.z, 1.0f),
          rect, gray);
  if (with_ceiling_light) {
    // the ceiling light the reference keeps commented out (src/scenes.rs:91-100)
    s.spawn(make_transform(dims.x * 0.2f, dims.z * 0.2f, 1.0f, 0.0f, -kFracPi2, 0.0f, 0.0f,
                           0.0f + pos.x, (dims.y / 2.0f - 0.01f) + pos.y, 0.0f + pos.z, 1.0f),
            rect, light);
  }
  // Back wall (src/scenes.rs:103-111)
  s.spawn(make_transform(dims.x, dims.y, 1.0f, 0.0f, kPi * 2.0f, 0.0f, 0.0f,
                         0.0f + pos.x, 0.0f + pos.y, dims.z / 2.0f + pos.z, 1.0f),
          rect, gray);
  // Red wall (src/scenes.rs:114-122)
  s.spawn(make_transform(dims.z, dims.y, 1.0f, 0.0f, 0.0f, kFracPi2, 0.0f,
                         dims.x / 2.0f + pos.x, 0.0f + pos.y, 0.0f + pos.z, 1.0f),
          rect, red);
  // Green wall (src/scenes.rs:125-133)
  s.spawn(make_transform(dims.z, dims.y, 1.0f, 0.0f, 0.0f, -kFracPi2, 0.0f,
                         -dims.x / 2.0f + pos.x, 0.0f + pos.y, 0.0f + pos.z, 1.0f),
          rect, green);
}

// ------------------------------------------------------- procedural "dragon"

namespace {

// sin and cos from +, -, * and comparisons on IEEE doubles only (this library is
// built without contraction), so that the procedural meshes come out with the
// same bits on every host whatever its libm: Cody-Waite reduction by pi/2 in
// two pieces (arguments here stay below a few thousand), then the Taylor
// polynomials on [-pi/4, pi/4], whose first neglected terms are below 1e-19.
void det_sincos(double x, double &s, double &c) {
  const double inv_pio2 = 0.63661977236758134308;
  const double pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
  const double kf = x * inv_pio2;
  const long long k = (long long)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
  const double kd = (double)k;
  const double r = (x - kd * pio2_hi) - kd * pio2_lo;
  const double r2 = r * r;
  double ps = -1.0 / 121645100408832000.0;  // r^19
  ps = ps * r2 + 1.0 / 355687428096000.0;
  ps = ps * r2 - 1.0 / 1307674368000.0;
  ps = ps * r2 + 1.0 / 6227020800.0;
  ps = ps * r2 - 1.0 / 39916800.0;
  ps = ps * r2 + 1.0 / 362880.0;
  ps = ps * r2 - 1.0 / 5040.0;
  ps = ps * r2 + 1.0 / 120.0;
  ps = ps * r2 - 1.0 / 6.0;
  const double sr = r + r * r2 * ps;
  double pc = 1.0 / 2432902008176640000.0;  // r^20
  pc = pc * r2 - 1.0 / 6402373705728000.0;
  pc = pc * r2 + 1.0 / 20922789888000.0;
  pc = pc * r2 - 1.0 / 87178291200.0;
  pc = pc * r2 + 1.0 / 479001600.0;
  pc = pc * r2 - 1.0 / 3628800.0;
  pc = pc * r2 + 1.0 / 40320.0;
  pc = pc * r2 - 1.0 / 720.0;
  pc = pc * r2 + 1.0 / 24.0;
  pc = pc * r2 - 0.5;
  const double cr = 1.0 + r2 * pc;
  switch ((int)(k & 3)) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
  }
}
double det_sin(double x) { double s, c; det_sincos(x, s, c); return s; }
double det_cos(double x) { double s, c; det_sincos(x, s, c); return c; }

}  // namespace

double host_det_sin(double x) { return det_sin(x); }
double host_det_cos(double x) { return det_cos(x); }

ObjModel make_knot_model(uint32_t n_u, uint32_t n_v, uint32_t seed) {
  // A (2,3) torus knot swept by a tube whose radius is displaced by a few
  // octaves of seeded sinusoids: a closed, bumpy, self-occluding surface with
  // 2*n_u*n_v triangles.  Only IEEE +, -, *, / and sqrt are used (det_sincos
  // above), so the mesh has the same bits on every host.
  ObjModel m;
  SplitMix64 rng(0xD8A60000ull + seed);
  const int kOct = 5;
  float fu[kOct], fv[kOct], ph[kOct], am[kOct];
  for (int o = 0; o < kOct; ++o) {
    fu[o] = (float)(3 + rng.below(6)) * (float)(1 << o);
    fv[o] = (float)(1 + rng.below(4)) * (float)(1 << o);
    ph[o] = rng.range(0.0f, 6.2831853f);
    am[o] = 0.35f / (float)(1 << o);
  }
  const double two_pi = 6.283185307179586;
  m.positions.reserve((size_t)n_u * n_v * 3);
  for (uint32_t i = 0; i < n_u; ++i) {
    double u = two_pi * (double)i / (double)n_u;
    // knot centre line and its frame
    auto curve = [](double t, double c[3]) {
      double r = 2.0 + det_cos(3.0 * t);
      c[0] = r * det_cos(2.0 * t);
      c[1] = r * det_sin(2.0 * t);
      c[2] = -det_sin(3.0 * t) * 1.5;
    };
    double c0[3], c1[3];
    curve(u, c0);
    curve(u + 1e-4, c1);
    double tx = c1[0] - c0[0], ty = c1[1] - c0[1], tz = c1[2] - c0[2];
    double tl = sqrt(tx * tx + ty * ty + tz * tz);
    tx /= tl; ty /= tl; tz /= tl;
    // normal = component of -c0 orthogonal to the tangent
    double nx = -c0[0], ny = -c0[1], nz = -c0[2];
    double d = nx * tx + ny * ty + nz * tz;
    nx -= d * tx; ny -= d * ty; nz -= d * tz;
    double nl = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= nl; ny /= nl; nz /= nl;
    double bx = ty * nz - tz * ny, by = tz * nx - tx * nz, bz = tx * ny - ty * nx;
    for (uint32_t j = 0; j < n_v; ++j) {
      double v = two_pi * (double)j / (double)n_v;
      double rad = 0.55;
      for (int o = 0; o < kOct; ++o) rad += 0.55 * am[o] * det_sin(fu[o] * u + fv[o] * v + ph[o]);
      double cx = det_cos(v) * rad, sx = det_sin(v) * rad;
      m.positions.push_back((float)(c0[0] + cx * nx + sx * bx));
      m.positions.push_back((float)(c0[1] + cx * ny + sx * by));
      m.positions.push_back((float)(c0[2] + cx * nz + sx * bz));
    }
  }
  m.indices.reserve((size_t)n_u * n_v * 6);
  for (uint32_t i = 0; i < n_u; ++i) {
    uint32_t i1 = (i + 1) % n_u;
    for (uint32_t j = 0; j < n_v; ++j) {
      uint32_t j1 = (j + 1) % n_v;
      uint32_t a = i * n_v + j, b = i1 * n_v + j, c = i1 * n_v + j1, dd = i * n_v + j1;
      uint32_t t[6] = {a, b, c, a, c, dd};
      m.indices.insert(m.indices.end(), t, t + 6);
    }
  }
  return m;
}

// --------------------------------------------- subdivided asset ("scan-like")

ObjModel subdivide_model(const ObjModel &in, uint32_t levels, uint32_t seed) {
  // Every level splits each triangle into four at its edge midpoints; a midpoint
  // is then moved by a seeded pseudo-random offset of at most +-4 % of its edge's
  // length per axis (scanner-like irregularity: no two centroids or box faces
  // coincide exactly).  File normals are dropped, so the mesh takes the
  // computed-normal path of Mesh::from_model.  Integer hashing and IEEE +, -, *
  // only: the same bits on every host.
  ObjModel m;
  m.positions = in.positions;
  m.indices = in.indices;
  auto hash32 = [](uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
  };
  for (uint32_t level = 0; level < levels; ++level) {
    std::unordered_map<uint64_t, uint32_t> mid;
    mid.reserve(m.indices.size());
    std::vector<uint32_t> out;
    out.reserve(m.indices.size() * 4);
    auto midpoint = [&](uint32_t a, uint32_t b) {
      const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = mid.find(key);
      if (it != mid.end()) return it->second;
      const uint32_t id = (uint32_t)(m.positions.size() / 3);
      const float *pa = &m.positions[(size_t)a * 3], *pb = &m.positions[(size_t)b * 3];
      const float ex = pb[0] - pa[0], ey = pb[1] - pa[1], ez = pb[2] - pa[2];
      const float len = sqrtf(ex * ex + ey * ey + ez * ez);
      float p[3];
      for (int k = 0; k < 3; ++k) {
        const uint32_t h = hash32(hash32(id * 3u + (uint32_t)k) ^ (seed * 0x9E3779B9u + level));
        const float j = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.08f;  // [-0.04, 0.04)
        p[k] = (pa[k] + pb[k]) * 0.5f + j * len;
      }
      m.positions.insert(m.positions.end(), p, p + 3);
      mid.emplace(key, id);
      return id;
    };
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3) {
      const uint32_t a = m.indices[t], b = m.indices[t + 1], c = m.indices[t + 2];
      const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      const uint32_t tri[12] = {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca};
      out.insert(out.end(), tri, tri + 12);
    }
    m.indices.swap(out);
  }
  return m;
}

// ----------------------------------------------------------- sample sources

std::vector<MirtSampleSource> make_sample_sources(uint32_t width, uint32_t height, uint64_t seed,
                                                  int shuffle) {
  // src/pathtracer_state.rs:175-195 shuffles 128x128 tiles and then all
  // sources with OS entropy and drops partial tiles.  Here partial tiles are
  // kept (every pixel is covered) and both shuffles are Fisher-Yates draws
  // from a seeded generator.  shuffle: 2 = both shuffles (the reference's
  // order), 1 = tiles only, 0 = none.
  const uint32_t tile = 128;
  SplitMix64 rng(seed);
  std::vector<std::pair<uint32_t, uint32_t>> tiles;
  for (uint32_t x = 0; x < (width + tile - 1) / tile; ++x)
    for (uint32_t y = 0; y < (height + tile - 1) / tile; ++y) tiles.push_back({x, y});
  if (shuffle >= 1)
    for (size_t i = tiles.size(); i > 1; --i) std::swap(tiles[i - 1], tiles[rng.next() % i]);
  std::vector<MirtSampleSource> data;
  data.reserve((size_t)width * height);
  for (auto [tx, ty] : tiles)
    for (uint32_t x = tx * tile; x < std::min(width, tx * tile + tile); ++x)
      for (uint32_t y = ty * tile; y < std::min(height, ty * tile + tile); ++y) {
        MirtSampleSource s;
        s.screen_pos[0] = (float)x / (float)width;
        s.screen_pos[1] = (float)y / (float)height;
        s.out_pos[0] = x;
        s.out_pos[1] = y;
        s.sample_count = 0;
        s.flags = 0;
        data.push_back(s);
      }
  if (shuffle >= 2)
    for (size_t i = data.size(); i > 1; --i) std::swap(data[i - 1], data[rng.next() % i]);
  return data;
}

}  // namespace mirth
