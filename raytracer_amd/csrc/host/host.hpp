// host.hpp — C++ restatement of the reference's host scene model.
//
// The reference keeps its scene model in Rust (src/mesh.rs, src/blas.rs,
// src/bvh.rs, src/tlas.rs, src/binder.rs, src/material.rs, src/camera.rs,
// src/scenes.rs).  No Rust toolchain exists where this library is built, so
// the same model is restated here in C++ with the same names and the same
// arithmetic (glam f32 semantics: lane-wise ops, sequential sums, true
// divisions).  It produces exactly the flat buffers the C ABI of libmirt
// consumes; it contains no GPU code.
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "mirt.h"
#include "mirt_host.h"

namespace mirth {

struct Vec3 {
  float x = 0, y = 0, z = 0;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct Vec4 {
  float x = 0, y = 0, z = 0, w = 0;
};

inline Vec4 operator+(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Vec4 operator-(Vec4 a, Vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Vec4 operator/(Vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float fmin2(float a, float b) { return b < a ? b : a; }  // SSE minps(a,b)
inline float fmax2(float a, float b) { return b > a ? b : a; }  // SSE maxps(a,b)
inline Vec4 vmin(Vec4 a, Vec4 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z), fmin2(a.w, b.w)}; }
inline Vec4 vmax(Vec4 a, Vec4 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z), fmax2(a.w, b.w)}; }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)}; }
inline Vec3 truncate(Vec4 a) { return {a.x, a.y, a.z}; }

// src/bvh.rs:5-18
struct AABB {
  Vec3 lb, ub;
  AABB unite(const AABB &o) const { return {vmin(lb, o.lb), vmax(ub, o.ub)}; }
};

// src/bvh.rs:20-29
struct BVHNode {
  AABB bounds;
  size_t left = 0, right = 0, skip = 0;
  bool is_leaf = false;
  size_t start = 0, end = 0;
};

// What tobj::Mesh holds for one model loaded with triangulate = true and
// single_index = false (src/mesh.rs:122-127): flat float/uint arrays.
struct ObjModel {
  std::vector<float> positions;   // 3 per vertex, in order of first use
  std::vector<float> normals;     // 3 per normal, in order of first use
  std::vector<uint32_t> indices;  // 3 per triangle, into positions
};

// Parse the first model of an OBJ file the way tobj 4.0.3 does with the
// reference's load options.  Returns false and sets `err` on failure.
bool load_obj_first_model(const std::string &path, ObjModel &out, std::string &err);
bool parse_obj_first_model(const char *text, size_t len, ObjModel &out, std::string &err);

// src/mesh.rs:24-30
struct Mesh {
  std::vector<Vec4> positions;
  std::vector<Vec4> normals;
  std::vector<uint32_t> faces;  // 4 per face (UVec4, w = 0)

  static Mesh from_model(const ObjModel &m);  // src/mesh.rs:308-355
  static Mesh rect();                         // src/mesh.rs:415-437
  static Mesh cube();                         // src/mesh.rs:439-536
  size_t n_faces() const { return faces.size() / 4; }
};
std::vector<Vec4> compute_vertex_normals_ccw(const std::vector<Vec4> &positions,
                                             const uint32_t *indices, size_t n_indices);

// src/blas.rs:12-79 — owns the mesh; building reorders mesh.faces in place.
struct BLAS {
  std::vector<BVHNode> nodes;
  Mesh mesh;
  explicit BLAS(Mesh m);
};

// src/mesh.rs:58-62
struct MeshData {
  std::vector<MirtBVHNode> nodes;
  Mesh mesh;
  AABB aabb;
  // texture coordinates, two per vertex; empty = all zero, which is what the
  // reference uploads for every mesh (src/mesh.rs:241)
  std::vector<float> uvs;
};

// src/tlas.rs:20-159
struct TLAS {
  std::vector<BVHNode> nodes;
  std::vector<size_t> instance_ids;
  std::vector<AABB> aabbs;
  TLAS(const std::vector<AABB> &geom_aabbs, const std::vector<MirtTransform> &transforms,
       const std::vector<MirtInstance> &instances);
};

MirtBVHNode to_gpu(const BVHNode &n);  // src/bvh.rs:212-224

// Deterministic procedural stand-in for a Stanford-dragon-class mesh: a
// (p,q) torus-knot tube with seeded multi-octave displacement, written as an
// ObjModel without normals (so Mesh::from_model computes them).
ObjModel make_knot_model(uint32_t n_u, uint32_t n_v, uint32_t seed);
// A loaded model with every triangle split into four `levels` times and the
// new vertices jittered by a seeded fraction of their edge (no file normals).
ObjModel subdivide_model(const ObjModel &in, uint32_t levels, uint32_t seed);
// sin / cos from IEEE +, -, * only (the procedural meshes use nothing else)
double host_det_sin(double x);
double host_det_cos(double x);

// ECS entity (Transform, MeshId, MaterialId) of src/binder.rs:52.
struct Entity {
  MirtTransform transform;
  uint32_t mesh_id;
  uint32_t material_id;
};

// MeshServer (src/mesh.rs:70-285) + MaterialServer (src/material.rs:47-71) +
// the entity list + the packing rules of binder_system (src/binder.rs:198-282).
// A BLAS builder supplied by the caller (the signature of mirt_build_blas
// without its context): reorders faces4 in place, fills `nodes`, returns 0 on
// success.  When set, add_mesh uses it instead of the host builder.
typedef int (*BlasBuilderFn)(void *user, const float *positions4, uint32_t n_positions, uint32_t *faces4,
                             uint32_t n_faces, MirtBVHNode *nodes, uint32_t nodes_capacity, uint32_t *n_nodes);

struct Scene {
  BlasBuilderFn blas_builder = nullptr;
  void *blas_builder_user = nullptr;
  std::string build_error;  // set when the callback fails; add_mesh then returns 0xFFFFFFFF
  // MeshServer
  std::vector<MeshData> meshes;
  std::unordered_map<std::string, uint32_t> by_desc;
  // MaterialServer
  std::vector<MirtMaterial> material_server;
  // textures the materials name by 1-based id (the reference has ids and no texture store, src/material.rs:15-18)
  std::vector<MirtTexture> textures;
  std::vector<uint32_t> texels;
  // entities in spawn order
  std::vector<Entity> entities;

  // flat buffers (valid after bind())
  std::vector<MirtVertex> vertices;
  std::vector<MirtIndex> indices;
  std::vector<MirtBVHNode> blas_nodes;
  std::vector<MirtGeometryOffsets> geometry_offsets;
  std::vector<AABB> aabbs;
  std::vector<MirtInstance> instances;
  std::vector<MirtMaterial> materials;
  std::vector<MirtTransform> transforms;
  std::vector<MirtBVHNode> tlas_nodes;
  std::vector<uint32_t> tlas_to_instances;
  std::vector<uint32_t> light_sources;
  // per-material kernels' tables derived from `materials`
  std::vector<MirtClassicMaterialRef> classic_refs;
  std::vector<MirtLambertianData> lambertian;
  std::vector<MirtMetallicData> metallic;
  std::vector<MirtDielectricData> dielectric;
  std::vector<MirtEmissiveData> emissive;

  uint32_t add_mesh(const std::string &desc, Mesh mesh);  // load_mesh + BLAS build
  uint32_t add_material(const MirtMaterial &m);
  uint32_t spawn(const MirtTransform &t, uint32_t mesh_id, uint32_t material_id);
  void regenerate_buffer();  // src/mesh.rs:198-284
  void bind();               // src/binder.rs:184-282
  void fill_desc(MirtSceneDesc &d) const;
};

// gltf.cpp: a glTF 2.0 file into meshes, materials, textures and entities of `scene`
bool load_gltf(Scene &scene, const std::string &path, MirthGltfInfo &info, std::string &err);
// PNG bytes -> RGBA8 texels (red in the low byte), rows top to bottom
bool decode_png_rgba8(const unsigned char *png, size_t len, uint32_t &width, uint32_t &height,
                      std::vector<uint32_t> &texels, std::string &err);

MirtMaterial default_material();  // src/material.rs:27-42
MirtCamera default_camera();      // src/camera.rs:117-128

// src/scenes.rs:20-134
void spawn_cornell(Scene &s, Vec3 dims, Vec3 pos, bool with_ceiling_light);

// Sample sources in the reference's tile-shuffled style
// (src/pathtracer_state.rs:175-195) but seeded and covering every pixel.
std::vector<MirtSampleSource> make_sample_sources(uint32_t width, uint32_t height, uint64_t seed,
                                                  int shuffle);

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform float in [0,1) with 24 bits
  float unit() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
  float range(float lo, float hi) { return lo + (hi - lo) * unit(); }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

}  // namespace mirth
