// mesh_bvh.cpp — Mesh normalisation, the median-split BVH builder with skip
// links, and its BLAS / TLAS adaptors.
//
// Follows src/mesh.rs:287-537, src/bvh.rs:37-224, src/blas.rs:18-79 and
// src/tlas.rs:27-159, including the behaviours that decide triangle ids and
// node order: the partition loop that never tests the element where the two
// cursors meet, the centroid that adds positions[0] as a fourth term, the
// anisotropic normalisation, children pushed left-then-right before the
// left-first recursion, and skip links threaded left -> right -> parent's next.
#include <cmath>
#include <cstring>
#include <functional>
#include <pthread.h>

#include "host.hpp"

namespace mirth {

// --------------------------------------------------------------------- Mesh

std::vector<Vec4> compute_vertex_normals_ccw(const std::vector<Vec4> &positions,
                                             const uint32_t *indices, size_t n_indices) {
  // src/mesh.rs:357-413
  std::vector<Vec4> acc(positions.size());
  for (size_t t = 0; t + 2 < n_indices; t += 3) {
    size_t i0 = indices[t], i1 = indices[t + 1], i2 = indices[t + 2];
    Vec4 p0 = positions[i0], p1 = positions[i1], p2 = positions[i2];
    float e1[3] = {p1.x - p0.x, p1.y - p0.y, p1.z - p0.z};
    float e2[3] = {p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2],
                  e1[0] * e2[1] - e1[1] * e2[0]};
    float l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    if (l2 > 0.0f) {
      float inv_len = 1.0f / sqrtf(l2);
      n[0] *= inv_len;
      n[1] *= inv_len;
      n[2] *= inv_len;
      for (size_t i : {i0, i1, i2}) {
        acc[i].x += n[0];
        acc[i].y += n[1];
        acc[i].z += n[2];
      }
    }
  }
  for (Vec4 &a : acc) {
    float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
    if (l2 > 0.0f) {
      float inv_len = 1.0f / sqrtf(l2);
      a.x *= inv_len;
      a.y *= inv_len;
      a.z *= inv_len;
    } else {
      a.z = 1.0f;
    }
    a.w = 0.0f;
  }
  return acc;
}

Mesh Mesh::from_model(const ObjModel &model) {
  // src/mesh.rs:308-355
  const size_t len = model.positions.size() / 3;
  std::vector<Vec4> positions(len);
  for (size_t i = 0; i < len; ++i)
    positions[i] = {model.positions[3 * i], model.positions[3 * i + 1], model.positions[3 * i + 2], 0.0f};

  Vec4 sum{0, 0, 0, 0};  // iter().sum() is a left fold from ZERO
  for (const Vec4 &p : positions) sum = sum + p;
  const Vec4 center = sum / (float)len;
  for (Vec4 &p : positions) p = p - center;

  Vec4 extent{0, 0, 0, 0};
  if (len > 0) {
    extent = positions[0];
    for (size_t i = 1; i < len; ++i) extent = vmax(positions[i], extent);
  }
  for (Vec4 &p : positions) p = {p.x / extent.x, p.y / extent.y, p.z / extent.z, 1.0f};

  Mesh m;
  m.faces.reserve(model.indices.size() / 3 * 4);
  for (size_t t = 0; t + 2 < model.indices.size(); t += 3) {
    m.faces.push_back(model.indices[t]);
    m.faces.push_back(model.indices[t + 1]);
    m.faces.push_back(model.indices[t + 2]);
    m.faces.push_back(0);
  }
  if (model.normals.size() >= model.positions.size() && !model.normals.empty()) {
    const size_t nn = model.normals.size() / 3;
    m.normals.resize(nn);
    for (size_t i = 0; i < nn; ++i)
      m.normals[i] = {model.normals[3 * i], model.normals[3 * i + 1], model.normals[3 * i + 2], 0.0f};
  } else {
    m.normals = compute_vertex_normals_ccw(positions, model.indices.data(), model.indices.size());
  }
  m.positions = std::move(positions);
  return m;
}

Mesh Mesh::rect() {
  // src/mesh.rs:415-437
  Mesh m;
  m.positions = {{-0.5f, -0.5f, 0.0f, 1.0f}, {0.5f, -0.5f, 0.0f, 1.0f}, {0.5f, 0.5f, 0.0f, 1.0f}, {-0.5f, 0.5f, 0.0f, 1.0f}};
  m.normals.assign(4, Vec4{0.0f, 0.0f, 1.0f, 0.0f});
  m.faces = {0, 1, 2, 0, 0, 2, 3, 0};
  return m;
}

Mesh Mesh::cube() {
  // src/mesh.rs:439-536 — six faces of four corners each, listed +X,-X,+Y,-Y,+Z,-Z
  static const float P[24][3] = {
      {0.5f, -0.5f, -0.5f},  {0.5f, 0.5f, -0.5f},  {0.5f, 0.5f, 0.5f},   {0.5f, -0.5f, 0.5f},
      {-0.5f, -0.5f, 0.5f},  {-0.5f, 0.5f, 0.5f},  {-0.5f, 0.5f, -0.5f}, {-0.5f, -0.5f, -0.5f},
      {-0.5f, 0.5f, -0.5f},  {-0.5f, 0.5f, 0.5f},  {0.5f, 0.5f, 0.5f},   {0.5f, 0.5f, -0.5f},
      {-0.5f, -0.5f, 0.5f},  {-0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, 0.5f},
      {-0.5f, -0.5f, 0.5f},  {0.5f, -0.5f, 0.5f},  {0.5f, 0.5f, 0.5f},   {-0.5f, 0.5f, 0.5f},
      {0.5f, -0.5f, -0.5f},  {-0.5f, -0.5f, -0.5f}, {-0.5f, 0.5f, -0.5f}, {0.5f, 0.5f, -0.5f}};
  static const float N[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  Mesh m;
  for (int i = 0; i < 24; ++i) {
    m.positions.push_back({P[i][0], P[i][1], P[i][2], 1.0f});
    m.normals.push_back({N[i / 4][0], N[i / 4][1], N[i / 4][2], 0.0f});
  }
  for (uint32_t f = 0; f < 6; ++f) {
    uint32_t b = 4 * f;
    uint32_t tri[8] = {b, b + 1, b + 2, 0, b, b + 2, b + 3, 0};
    m.faces.insert(m.faces.end(), tri, tri + 8);
  }
  return m;
}

// ---------------------------------------------------------------------- BVH

namespace {

// The BVH trait of src/bvh.rs:37-179 over abstract elements.
struct Builder {
  std::vector<BVHNode> &nodes;
  std::function<AABB(size_t)> elem_bounds;
  std::function<Vec3(size_t)> elem_centroid;
  std::function<void(size_t, size_t)> elem_swap;

  void compute_node_bounds(size_t idx) {  // src/bvh.rs:52-66
    BVHNode node = nodes[idx];
    if (!node.is_leaf) {
      node.bounds = nodes[node.left].bounds.unite(nodes[node.right].bounds);
    } else {
      AABB b = elem_bounds(node.start);
      for (size_t i = node.start + 1; i < node.end; ++i) b = b.unite(elem_bounds(i));
      node.bounds = b;
    }
    nodes[idx] = node;
  }

  void subdivide(size_t idx, size_t threshold) {  // src/bvh.rs:68-157
    BVHNode node = nodes[idx];
    if (!node.is_leaf) {
      subdivide(node.left, threshold);
      subdivide(node.right, threshold);
      return;
    }
    if (node.end - node.start <= threshold) return;

    Vec3 extent = node.bounds.ub - node.bounds.lb;
    int axis = 0;
    if (extent.y > extent.x) axis = 1;
    if (extent.z > extent[axis]) axis = 2;

    size_t i = node.start, j = node.end - 1;
    Vec3 cmin = elem_centroid(node.start), cmax = cmin;
    for (size_t k = node.start + 1; k < node.end; ++k) {
      Vec3 c = elem_centroid(k);
      cmin = vmin(cmin, c);
      cmax = vmax(cmax, c);
    }
    const float split = (cmin[axis] + cmax[axis]) * 0.5f;

    while (i < j) {
      if (elem_centroid(i)[axis] < split) {
        i += 1;
      } else {
        elem_swap(i, j);
        j -= 1;
      }
    }
    if (i == node.end || i == node.start) return;

    BVHNode left, right;
    left.is_leaf = true;
    left.start = node.start;
    left.end = i;
    right.is_leaf = true;
    right.start = i;
    right.end = node.end;
    const size_t l = nodes.size();
    nodes.push_back(left);
    const size_t r = nodes.size();
    nodes.push_back(right);

    compute_node_bounds(l);
    compute_node_bounds(r);
    subdivide(l, threshold);
    subdivide(r, threshold);

    BVHNode inner;
    inner.is_leaf = false;
    inner.left = l;
    inner.right = r;
    inner.start = node.start;
    inner.end = node.end;
    nodes[idx] = inner;
    compute_node_bounds(idx);
  }

  void generate_skips(size_t idx, size_t next) {  // src/bvh.rs:159-172
    nodes[idx].skip = next;
    const BVHNode node = nodes[idx];
    if (!node.is_leaf) {
      generate_skips(node.left, node.right);
      generate_skips(node.right, next);
    }
  }

  void initialize(size_t threshold) {  // src/bvh.rs:174-178
    compute_node_bounds(0);
    subdivide(0, threshold);
    generate_skips(0, 0);
  }
};

// The recursion depth follows the tree depth, which degenerate inputs can
// make linear in the element count; run the build on a thread with a large
// stack so that it cannot overflow.
void run_with_big_stack(const std::function<void()> &fn) {
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, (size_t)1 << 30);
  pthread_t th;
  auto tramp = [](void *p) -> void * {
    (*static_cast<const std::function<void()> *>(p))();
    return nullptr;
  };
  if (pthread_create(&th, &attr, tramp, (void *)&fn) != 0) {
    fn();
  } else {
    pthread_join(th, nullptr);
  }
  pthread_attr_destroy(&attr);
}

}  // namespace

MirtBVHNode to_gpu(const BVHNode &n) {  // src/bvh.rs:190-224
  MirtBVHNode g;
  memset(&g, 0, sizeof(g));
  g.lb[0] = n.bounds.lb.x;
  g.lb[1] = n.bounds.lb.y;
  g.lb[2] = n.bounds.lb.z;
  g.ub[0] = n.bounds.ub.x;
  g.ub[1] = n.bounds.ub.y;
  g.ub[2] = n.bounds.ub.z;
  g.left = (uint32_t)n.left;
  g.right = (uint32_t)n.skip;
  g.is_leaf = n.is_leaf ? 1u : 0u;
  g.start = (uint32_t)n.start;
  g.end = (uint32_t)n.end;
  return g;
}

BLAS::BLAS(Mesh m) : mesh(std::move(m)) {  // src/blas.rs:62-78
  BVHNode root;
  root.is_leaf = true;
  root.start = 0;
  root.end = mesh.n_faces();
  nodes.push_back(root);
  if (mesh.n_faces() == 0) return;

  Builder b{nodes,
            // src/blas.rs:19-31
            [this](size_t f) {
              const uint32_t *face = &mesh.faces[4 * f];
              Vec4 p0 = mesh.positions[face[0]], p1 = mesh.positions[face[1]], p2 = mesh.positions[face[2]];
              return AABB{truncate(vmin(vmin(p0, p1), p2)), truncate(vmax(vmax(p0, p1), p2))};
            },
            // src/blas.rs:33-37 — face.to_array() has four entries, the fourth is 0
            [this](size_t f) {
              const uint32_t *face = &mesh.faces[4 * f];
              Vec4 acc = mesh.positions[face[0]];
              acc = acc + mesh.positions[face[1]];
              acc = acc + mesh.positions[face[2]];
              acc = acc + mesh.positions[face[3]];
              return truncate(acc / 3.0f);
            },
            // src/blas.rs:39-41
            [this](size_t a, size_t c) {
              for (int k = 0; k < 4; ++k) std::swap(mesh.faces[4 * a + k], mesh.faces[4 * c + k]);
            }};
  run_with_big_stack([&] { b.initialize(4); });
}

namespace {
struct Mat4 {  // glam column-major
  float c[4][4];
};
Mat4 mat_identity() {
  Mat4 m;
  memset(&m, 0, sizeof(m));
  m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
  return m;
}
void mul_vec4(const Mat4 &m, const float v[4], float out[4]) {
  // x*col0 + y*col1 + z*col2 + w*col3, summed left to right
  for (int r = 0; r < 4; ++r) {
    float acc = m.c[0][r] * v[0];
    acc = acc + m.c[1][r] * v[1];
    acc = acc + m.c[2][r] * v[2];
    acc = acc + m.c[3][r] * v[3];
    out[r] = acc;
  }
}
Mat4 mul_mat4(const Mat4 &a, const Mat4 &b) {
  Mat4 o;
  for (int j = 0; j < 4; ++j) mul_vec4(a, b.c[j], o.c[j]);
  return o;
}
}  // namespace

TLAS::TLAS(const std::vector<AABB> &geom_aabbs, const std::vector<MirtTransform> &transforms,
           const std::vector<MirtInstance> &instances) {
  // src/tlas.rs:61-99
  for (const MirtInstance &inst : instances) {
    const AABB &a = geom_aabbs[inst.geometry];
    const MirtTransform &t = transforms[inst.transform];

    Mat4 translate = mat_identity();
    translate.c[3][0] = t.translation[0];
    translate.c[3][1] = t.translation[1];
    translate.c[3][2] = t.translation[2];

    float sx = sinf(t.rotation[0]), cx = cosf(t.rotation[0]);
    float sy = sinf(t.rotation[1]), cy = cosf(t.rotation[1]);
    float sz = sinf(t.rotation[2]), cz = cosf(t.rotation[2]);
    Mat4 rx = mat_identity(), ry = mat_identity(), rz = mat_identity();
    rx.c[1][1] = cx;  rx.c[1][2] = sx;  rx.c[2][1] = -sx;  rx.c[2][2] = cx;
    ry.c[0][0] = cy;  ry.c[0][2] = -sy; ry.c[2][0] = sy;   ry.c[2][2] = cy;
    rz.c[0][0] = cz;  rz.c[0][1] = sz;  rz.c[1][0] = -sz;  rz.c[1][1] = cz;
    Mat4 rotate = mul_mat4(rx, mul_mat4(ry, rz));

    Mat4 scale = mat_identity();
    scale.c[0][0] = t.scale[0];
    scale.c[1][1] = t.scale[1];
    scale.c[2][2] = t.scale[2];

    Mat4 m = mul_mat4(translate, mul_mat4(rotate, scale));

    AABB acc;
    bool first = true;
    for (int x = 0; x <= 1; ++x)
      for (int y = 0; y <= 1; ++y)
        for (int z = 0; z <= 1; ++z) {
          float corner[4] = {x == 0 ? a.lb.x : a.ub.x, y == 0 ? a.lb.y : a.ub.y, z == 0 ? a.lb.z : a.ub.z, 1.0f};
          float o[4];
          mul_vec4(m, corner, o);
          AABB c{{o[0], o[1], o[2]}, {o[0], o[1], o[2]}};
          acc = first ? c : acc.unite(c);
          first = false;
        }
    aabbs.push_back(acc);
  }

  // src/tlas.rs:101-158
  BVHNode root;
  root.is_leaf = true;
  root.start = 0;
  root.end = instances.size();
  nodes.push_back(root);
  for (size_t i = 0; i < instances.size(); ++i) instance_ids.push_back(i);
  if (instances.empty()) return;

  Builder b{nodes,
            [this](size_t e) { return aabbs[e]; },                                // src/tlas.rs:28-30
            [this](size_t e) { return (aabbs[e].lb + aabbs[e].ub) / 2.0f; },      // src/tlas.rs:32-34
            [this](size_t a, size_t c) {                                          // src/tlas.rs:36-39
              std::swap(aabbs[a], aabbs[c]);
              std::swap(instance_ids[a], instance_ids[c]);
            }};
  run_with_big_stack([&] { b.initialize(1); });
}

}  // namespace mirth
