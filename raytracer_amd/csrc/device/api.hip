// api.hip — the extern "C" boundary of libmirt (include/mirt.h): context,
// scene upload into the traversal layout, path-state allocation, the frame
// loop, read-back and the ray-set entry points.
#include <math.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "kernels.h"
#include "mirt.h"

using namespace mirt;

// ---- the workspace of mirt_trace_closest_static: one file-scope array per
// device, so that the call allocates nothing.  Layout: 256 bytes of counters
// (five visit counters, the launch's chunk cursor), then kStaticTraceMaxRays
// rays (32 B each), then as many results (16 B each).
constexpr size_t kStaticTraceMaxRays = 131072;
constexpr size_t kStaticTraceRayOff = 256;
constexpr size_t kStaticTraceOutOff = kStaticTraceRayOff + kStaticTraceMaxRays * sizeof(MirtRay);
constexpr size_t kStaticTraceBytes = kStaticTraceOutOff + kStaticTraceMaxRays * sizeof(MirtHitId);
static __device__ __align__(256) unsigned char scratch[kStaticTraceBytes];

namespace {

// `scratch` names a host-side stand-in; the array's address on a device comes
// from hipGetSymbolAddress with that device current, once per device.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char *g_scratch_ptr[kMaxDevices] = {};
// One user of a device's array at a time within this process: the call is synchronous from its first
// write to the array to its last read, so holding this costs nothing and two contexts on one GPU used from
// two threads take turns instead of exchanging rays and results.
std::mutex g_scratch_use[kMaxDevices];

// With `device` current.  Returns nullptr when the address cannot be had.
unsigned char *static_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    g_scratch_ptr[device] = static_cast<unsigned char *>(p);
  }
  return g_scratch_ptr[device];
}

struct EventPair {
  hipEvent_t a, b;
  int kind;
};

enum KernelKind { K_SAMPLE = 0, K_EXTEND, K_FINISH, K_CONNECT, K_SHADE, K_KINDS };

}  // namespace

struct mirt_ctx {
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  std::string error;

  // scene
  bool have_scene = false;
  bool have_classic_tables = false;  // the scene came with classic_refs (per-material shading can run)
  DScene sc{};
  std::vector<void *> scene_allocs;     // geometry: nodes + triangles, normals
  std::vector<void *> instance_allocs;  // what mirt_scene_update_instances replaces
  uint32_t n_instances = 0, n_geometries = 0;
  uint32_t tlas_capacity = 0;  // nodes reserved for the TLAS at the front of the node array
  std::vector<uint32_t> geom_node_base, geom_tri_base, geom_root_start, geom_root_count;
  // kept for mirt_scene_update_transforms: each geometry's root box, the instances and their device records
  std::vector<float> geom_aabbs6;
  std::vector<MirtInstance> host_instances;
  std::vector<DInst> host_insts;
  uint32_t n_transforms = 0;
  uint32_t n_tlas_slots = 0;  // slots the uploaded TLAS lists (n_tlas_to_instances of the last upload)

  // state
  bool have_state = false;
  DState st{};
  std::vector<void *> state_allocs;
  MirtParams params{};
  MirtCamera camera{};
  bool have_camera = false;
  uint32_t next_sample = 0;  // sample number the next render call starts at
  uint32_t frame = 0;
  bool slots_seeded = false;
  bool rng_seeded = false;   // per-slot streams: mirt_state_set_rng has been called
  uint32_t stamp = 0;        // per-slot streams: last value written to DState::slot_stamp
  uint32_t *d_active = nullptr;  // the list DState::active points at (owned by state_allocs)
  uint32_t *d_export4 = nullptr; // staging for mirt_read_accum: the sums as 32-bit words (owned by state_allocs)

  // diagnostics
  bool counting = false;
  bool event_timing = false;
  std::vector<EventPair> ev_free, ev_pending;
  double kernel_ms[K_KINDS] = {0, 0, 0, 0, 0};
  uint64_t kernel_launches[K_KINDS] = {0, 0, 0, 0, 0};
  uint64_t frames_total = 0;
  unsigned long long *h_stats = nullptr;  // pinned, S_COUNT + Q_COUNT words
  unsigned long long *d_trace_counts = nullptr;
  unsigned char *static_ws = nullptr;  // this device's copy of the file-scope workspace (mirt_trace_closest_static)
  void *build_ws = nullptr;  // workspace of mirt_build_blas, grown on demand
  size_t build_ws_bytes = 0;
};

namespace {

int fail(mirt_ctx *c, int status, const std::string &msg) {
  if (c) c->error = msg;
  return status;
}

// No exception crosses the C boundary: every entry point is a function-try-block that ends here.  Host memory
// running out (std::bad_alloc from a staging vector) reads as MIRT_ERR_OUT_OF_MEMORY, anything else as
// MIRT_ERR_INTERNAL; the message goes to the context when one can still be written.
int exception_status(std::string &msg) noexcept {
  int status = MIRT_ERR_INTERNAL;
  try {
    try {
      throw;
    } catch (const std::bad_alloc &) {
      status = MIRT_ERR_OUT_OF_MEMORY;
      msg = "out of host memory";
    } catch (const std::exception &e) {
      msg = std::string("internal error: ") + e.what();
    } catch (...) {
      msg = "internal error";
    }
  } catch (...) {  // not even the message could be built
  }
  return status;
}
int on_exception(mirt_ctx *c) noexcept {
  std::string msg;
  const int status = exception_status(msg);
  try {
    if (c) c->error = msg;
  } catch (...) {
  }
  return status;
}

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail(ctx, _e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_HIP,       \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                           \
  } while (0)

// Every copy and fill of the context's memory goes through its stream: the stream is
// non-blocking, so work on the null stream (hipMemcpy, hipMemset) is not ordered with its
// kernels.  Complete on return, so the host side may be a temporary.
hipError_t copy_sync(hipStream_t stream, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, stream);
  return e == hipSuccess ? hipStreamSynchronize(stream) : e;
}

template <typename T>
int dev_alloc(mirt_ctx *c, std::vector<void *> &owner, T **out, size_t count) {
  void *p = nullptr;
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = 16;
  HIP_TRY(c, hipMalloc(&p, bytes));
  owner.push_back(p);
  *out = reinterpret_cast<T *>(p);
  return MIRT_OK;
}

template <typename T>
int dev_upload(mirt_ctx *c, std::vector<void *> &owner, const T **out, const std::vector<T> &host) {
  T *p = nullptr;
  int rc = dev_alloc(c, owner, &p, host.size());
  if (rc != MIRT_OK) return rc;
  if (!host.empty()) HIP_TRY(c, copy_sync(c->stream, p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return MIRT_OK;
}

void free_all(std::vector<void *> &v) {
  for (void *p : v) (void)hipFree(p);
  v.clear();
}

// ---- instance matrices: Transform.matrix() / matrix_inverse()
// (shaders/common.slang:41-111) evaluated once per instance on the host with
// the shader's own order of 4x4 products, instead of per ray x instance visit.
void mat4_mul(const float a[4][4], const float b[4][4], float o[4][4]) {
  float t[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = a[i][0] * b[0][j];
      acc = acc + a[i][1] * b[1][j];
      acc = acc + a[i][2] * b[2][j];
      acc = acc + a[i][3] * b[3][j];
      t[i][j] = acc;
    }
  memcpy(o, t, sizeof(t));
}
void mat4_identity(float m[4][4]) {
  memset(m, 0, 64);
  m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1.0f;
}
void instance_matrices(const MirtTransform &t, float M[12], float Mi[12]) {
  float S[4][4], R[4][4], T[4][4], rx[4][4], ry[4][4], rz[4][4], tmp[4][4], m[4][4];
  const float x = t.rotation[0], y = t.rotation[1], z = t.rotation[2];
  mat4_identity(rx); mat4_identity(ry); mat4_identity(rz);
  rx[1][1] = cosf(x); rx[1][2] = -sinf(x); rx[2][1] = sinf(x); rx[2][2] = cosf(x);
  ry[0][0] = cosf(y); ry[0][2] = sinf(y); ry[2][0] = -sinf(y); ry[2][2] = cosf(y);
  rz[0][0] = cosf(z); rz[0][1] = -sinf(z); rz[1][0] = sinf(z); rz[1][1] = cosf(z);
  mat4_mul(ry, rz, tmp);
  mat4_mul(rx, tmp, R);
  mat4_identity(S); S[0][0] = t.scale[0]; S[1][1] = t.scale[1]; S[2][2] = t.scale[2];
  mat4_identity(T); T[0][3] = t.translation[0]; T[1][3] = t.translation[1]; T[2][3] = t.translation[2];
  mat4_mul(R, S, tmp);
  mat4_mul(T, tmp, m);
  memcpy(M, m, 48);
  float Si[4][4], Rt[4][4], Ti[4][4], mi[4][4];
  mat4_identity(Si);
  Si[0][0] = 1.0f / t.scale[0]; Si[1][1] = 1.0f / t.scale[1]; Si[2][2] = 1.0f / t.scale[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Rt[i][j] = R[j][i];
  mat4_identity(Ti); Ti[0][3] = -t.translation[0]; Ti[1][3] = -t.translation[1]; Ti[2][3] = -t.translation[2];
  mat4_mul(Rt, Ti, tmp);
  mat4_mul(Si, tmp, mi);
  memcpy(Mi, mi, 48);
}

// fn(i) for i in [0, n) on up to sixteen host threads (the instance matrices: six sin / cos and
// four 4x4 products each; 200 000 instances take 15 ms on one core).  fn must not throw and must
// touch only element i.
template <class F>
void parallel_for(uint32_t n, F fn) {
  unsigned workers = std::thread::hardware_concurrency();
  if (workers > 16u) workers = 16u;
  if (workers < 2u || n < 4096u) {
    for (uint32_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  const uint32_t chunk = (n + workers - 1u) / workers;
  for (unsigned w = 0; w < workers; ++w) {
    const uint32_t lo = w * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) break;
    try {
      pool.emplace_back([=]() {
        for (uint32_t i = lo; i < hi; ++i) fn(i);
      });
    } catch (...) {  // no more threads to be had: this one does the slice itself
      for (uint32_t i = lo; i < hi; ++i) fn(i);
    }
  }
  for (std::thread &t : pool) t.join();
}

// The threaded walk moves from a node to its left child or to its skip link
// and ends when it arrives back at node 0.  It ends for every pattern of box
// hits iff the graph of those two kinds of edges, with the edges into node 0
// removed, has no cycle reachable from node 0.
bool walk_terminates(const MirtBVHNode *nodes, uint32_t n) {
  std::vector<uint8_t> colour(n, 0);  // 0 new, 1 on the DFS stack, 2 done
  std::vector<std::pair<uint32_t, uint8_t>> stack;
  stack.push_back({0u, 0});
  colour[0] = 1;
  while (!stack.empty()) {
    auto &[node, next_edge] = stack.back();
    const MirtBVHNode &nd = nodes[node];
    uint32_t target = 0;
    bool have = false;
    while (next_edge < 2 && !have) {
      if (next_edge == 0) { if (nd.is_leaf == 0) { target = nd.left; have = true; } }
      else { target = nd.right; have = true; }
      ++next_edge;
    }
    if (!have) {
      colour[node] = 2;
      stack.pop_back();
      continue;
    }
    if (target == 0u) continue;  // the walk ends there
    if (colour[target] == 1) return false;
    if (colour[target] == 0) {
      colour[target] = 1;
      stack.push_back({target, 0});
    }
  }
  return true;
}

// Every index a kernel will follow is checked here, once, so that no launch
// can read outside its buffers or walk forever.
// The instance-dependent part: instances, transforms, materials, TLAS, lights,
// per-material tables.
bool validate_instances(const MirtSceneDesc *d, std::string &why) {
  auto bad = [&](const char *m) { why = m; return false; };
  if (!d->instances || !d->n_instances) return bad("no instances");
  if (!d->n_geometries) return bad("missing geometry buffers");
  if (!d->materials || !d->n_materials || !d->transforms || !d->n_transforms) return bad("missing materials/transforms");
  if (!d->tlas_nodes || !d->n_tlas_nodes || !d->tlas_to_instances) return bad("missing TLAS buffers");
  if (d->n_tlas_nodes > kMaxNodesPerTree) return bad("TLAS has more than 2^24 nodes");
  for (uint32_t i = 0; i < d->n_instances; ++i)
    if (d->instances[i].transform >= d->n_transforms || d->instances[i].geometry >= d->n_geometries ||
        d->instances[i].material >= d->n_materials)
      return bad("instance refers to a missing transform, geometry or material");
  for (uint32_t i = 0; i < d->n_tlas_to_instances; ++i)
    if (d->tlas_to_instances[i] >= d->n_instances) return bad("tlas_to_instances entry out of range");
  for (uint32_t i = 0; i < d->n_tlas_nodes; ++i) {
    const MirtBVHNode &n = d->tlas_nodes[i];
    if (n.right >= d->n_tlas_nodes) return bad("TLAS skip link out of range");
    if (n.is_leaf > 1) return bad("TLAS is_leaf must be 0 or 1");
    if (n.is_leaf) {
      if (n.start >= n.end || n.end > d->n_tlas_to_instances) return bad("TLAS leaf empty or out of range");
    } else if (n.left >= d->n_tlas_nodes || n.left == 0) {
      return bad("TLAS child link out of range");
    }
  }
  if (!walk_terminates(d->tlas_nodes, d->n_tlas_nodes)) return bad("TLAS links form a cycle");
  if (d->n_light_sources != 0 && !d->light_sources) return bad("light_sources is null");
  const bool no_lights = d->n_light_sources == 0 || (d->n_light_sources == 1 && d->light_sources[0] == 0xFFFFFFFFu);
  if (!no_lights)
    for (uint32_t i = 0; i < d->n_light_sources; ++i)
      if (d->light_sources[i] >= d->n_instances) return bad("light source instance out of range");
  if (d->n_textures) {
    if (!d->textures || !d->texels) return bad("textures without texels");
    for (uint32_t i = 0; i < d->n_textures; ++i) {
      const MirtTexture &t = d->textures[i];
      if (t.width == 0 || t.height == 0 || t.width > 65536u || t.height > 65536u) return bad("texture size must be 1..65536");
      if (t.filter > MIRT_FILTER_BILINEAR) return bad("unknown texture filter");
      if ((unsigned long long)t.first_texel + (unsigned long long)t.width * t.height > d->n_texels)
        return bad("texture texels out of range");
    }
  }
  for (uint32_t i = 0; i < d->n_materials; ++i)
    if (d->materials[i].colour_texture > d->n_textures || d->materials[i].emissive_texture > d->n_textures)
      return bad("material names a texture the scene does not have");
  if (d->n_classic_refs) {
    if (d->n_classic_refs != d->n_materials || !d->classic_refs) return bad("classic_refs must have one entry per material");
    for (uint32_t i = 0; i < d->n_classic_refs; ++i) {
      const MirtClassicMaterialRef r = d->classic_refs[i];
      uint32_t lim = 0;
      switch (r.kind) {
        case MIRT_MAT_NONE: continue;
        case MIRT_MAT_LAMBERTIAN: lim = d->lambertian ? d->n_lambertian : 0; break;
        case MIRT_MAT_METALLIC: lim = d->metallic ? d->n_metallic : 0; break;
        case MIRT_MAT_DIELECTRIC: lim = d->dielectric ? d->n_dielectric : 0; break;
        case MIRT_MAT_EMISSIVE: lim = d->emissive ? d->n_emissive : 0; break;
        default: return bad("unknown classic material kind");
      }
      if (r.data >= lim) return bad("classic material row out of range");
    }
  }
  return true;
}

// The geometry part: per-geometry BLAS nodes, index rows, vertices.
bool validate_geometry(const MirtSceneDesc *d, std::string &why) {
  auto bad = [&](const char *m) { why = m; return false; };
  if (!d->geometry_offsets || !d->n_geometries || !d->vertices || !d->indices || !d->blas_nodes)
    return bad("missing geometry buffers");
  for (uint32_t g = 0; g < d->n_geometries; ++g) {
    const MirtGeometryOffsets &o = d->geometry_offsets[g];
    const bool last = g + 1 == d->n_geometries;
    const uint32_t node_end = last ? d->n_blas_nodes : d->geometry_offsets[g + 1].blas_node;
    const uint32_t idx_end = last ? d->n_indices : d->geometry_offsets[g + 1].index;
    const uint32_t vtx_end = last ? d->n_vertices : d->geometry_offsets[g + 1].vertex;
    if (o.blas_node >= node_end || node_end > d->n_blas_nodes || o.index > idx_end || idx_end > d->n_indices ||
        o.vertex > vtx_end || vtx_end > d->n_vertices)
      return bad("geometry offsets are not ascending or exceed the buffers");
    const uint32_t nn = node_end - o.blas_node, ni = idx_end - o.index, nv = vtx_end - o.vertex;
    if (nn > kMaxNodesPerTree) return bad("a BLAS has more than 2^24 nodes");
    if (ni > kMaxNodesPerTree) return bad("a geometry has more than 2^24 triangles");
    if (ni == 0) return bad("a geometry has no triangles");
    // Leaves must not share triangles: the first triangle of a big leaf carries
    // that leaf's size and the offsets of its scan blocks and LDS image.
    std::vector<uint8_t> in_leaf(ni, 0);
    for (uint32_t i = 0; i < nn; ++i) {
      const MirtBVHNode &n = d->blas_nodes[o.blas_node + i];
      if (n.right >= nn) return bad("BLAS skip link out of range");
      if (n.is_leaf > 1) return bad("BLAS is_leaf must be 0 or 1");
      if (n.is_leaf) {
        if (n.start >= n.end || n.end > ni) return bad("BLAS leaf empty or out of range");
        for (uint32_t p = n.start; p < n.end; ++p) {
          if (in_leaf[p]) return bad("BLAS leaves overlap");
          in_leaf[p] = 1;
        }
      } else if (n.left >= nn || n.left == 0) {
        return bad("BLAS child link out of range");
      }
    }
    if (!walk_terminates(d->blas_nodes + o.blas_node, nn)) return bad("BLAS links form a cycle");
    for (uint32_t i = 0; i < ni; ++i)
      for (int k = 0; k < 3; ++k)
        if (d->indices[o.index + i].v[k] >= nv) return bad("vertex index out of range");
  }
  return true;
}

bool validate_scene(const MirtSceneDesc *d, std::string &why) {
  return validate_geometry(d, why) && validate_instances(d, why);
}

// Pack one tree's reference nodes into DNodes.  `sizes_out` receives the
// out-of-line size of every node (only read for leaves marked kBigLeaf).
// TLAS leaves are all marked kBigLeaf (`all_big`): the walk hands every leaf
// so marked to its service point, which is where instances are entered.
void pack_nodes(const MirtBVHNode *src, uint32_t n, DNode *dst, uint32_t *sizes_out, bool all_big) {
  for (uint32_t i = 0; i < n; ++i) {
    const MirtBVHNode &s = src[i];
    DNode d;
    d.lbx = s.lb[0]; d.lby = s.lb[1]; d.lbz = s.lb[2];
    d.ubx = s.ub[0]; d.uby = s.ub[1]; d.ubz = s.ub[2];
    // is_leaf == 1 marks a leaf exactly like the kernels' `node.is_leaf == 1`
    // test; any other value walks on as an inner node.
    if (s.is_leaf == 1) {
      const uint32_t cnt = s.end - s.start;  // >= 1, checked by validate_scene
      const uint32_t tag = (cnt < kBigLeaf && !all_big) ? cnt : kBigLeaf;
      d.a = s.start;
      d.b = (s.right & kSkipMask) | (tag << 24);
      if (sizes_out) sizes_out[i] = cnt;
    } else {
      d.a = s.left;
      d.b = (s.right & kSkipMask);
      if (sizes_out) sizes_out[i] = 0;
    }
    dst[i] = d;
  }
}

int get_ctx_device(mirt_ctx *c) {
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, MIRT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return MIRT_OK;
}

#define ENTER(ctx)                                   \
  if (!(ctx)) return MIRT_ERR_INVALID_ARGUMENT;      \
  {                                                  \
    int _rc = get_ctx_device(ctx);                   \
    if (_rc != MIRT_OK) return _rc;                  \
  }

LaunchCfg launch_cfg(mirt_ctx *c) {
  LaunchCfg cfg;
  cfg.stream = c->stream;
  uint32_t want = c->have_state ? (c->st.slots + kStreamBlock - 1u) / kStreamBlock : 1u;
#ifndef MIRT_STREAM_BLOCKS_PER_CU
#define MIRT_STREAM_BLOCKS_PER_CU 2u
#endif
  uint32_t cap = (uint32_t)c->num_cus * MIRT_STREAM_BLOCKS_PER_CU;  // two 1024-thread blocks fill a CU's 32 wave slots
  cfg.grid_blocks = (int)(want < cap ? (want ? want : 1u) : cap);
  cfg.num_cus = c->num_cus;
  return cfg;
}

DFrameParams frame_params(mirt_ctx *c, unsigned long long sample_limit, uint32_t first_sample) {
  DFrameParams fp;
  memset(&fp, 0, sizeof(fp));
  fp.p = c->params;
  fp.cam = c->camera;
  fp.sample_limit = sample_limit;
  fp.first_sample = first_sample;
  fp.frame = c->frame;
  fp.have_lights = c->sc.n_lights > 0 ? 1u : 0u;
  fp.counting = c->counting ? 1u : 0u;
  fp.stamp = c->stamp;
  return fp;
}

// --- optional per-kernel timing with HIP events on the context's stream
void ev_begin(mirt_ctx *c, int kind, EventPair &ep) {
  if (c->ev_free.empty()) {
    EventPair n;
    (void)hipEventCreate(&n.a);
    (void)hipEventCreate(&n.b);
    c->ev_free.push_back(n);
  }
  ep = c->ev_free.back();
  c->ev_free.pop_back();
  ep.kind = kind;
  (void)hipEventRecord(ep.a, c->stream);
}
void ev_end(mirt_ctx *c, EventPair &ep) {
  (void)hipEventRecord(ep.b, c->stream);
  c->ev_pending.push_back(ep);
}
void ev_collect(mirt_ctx *c) {  // call after a stream synchronisation
  for (EventPair &ep : c->ev_pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) c->kernel_ms[ep.kind] += (double)ms;
    c->ev_free.push_back(ep);
  }
  c->ev_pending.clear();
}

template <typename F>
void timed_launch(mirt_ctx *c, int kind, F &&f) {
  c->kernel_launches[kind]++;
  if (c->event_timing) {
    EventPair ep;
    ev_begin(c, kind, ep);
    f();
    ev_end(c, ep);
  } else {
    f();
  }
}

int seed_slots(mirt_ctx *c) {
  launch_init_slots(launch_cfg(c), c->st);
  HIP_TRY(c, hipMemsetAsync(c->st.sample_index, 0, sizeof(unsigned long long), c->stream));
  c->frame = 0;
  c->slots_seeded = true;
  return MIRT_OK;
}

// `respawn`: the kernels in which paths end (the shade kernel with uber
// shading; k_extend_finish and the material kernel with per-material shading)
// accumulate them and respawn their slots at once.  The sample kernel then finds
// an empty terminate queue after the first frame and only clears the idle queue
// counters — except with MirtParams.stage_kernels, where it is left with the
// paths that end in one of the four material kernels.
void run_frame(mirt_ctx *c, unsigned long long sample_limit, uint32_t first_sample, bool respawn) {
  const LaunchCfg cfg = launch_cfg(c);
  const bool per_slot = c->params.rng_mode == MIRT_RNG_PER_SLOT;
  if (per_slot) {
    respawn = false;  // the sample stage alone pairs slots with sample numbers, in slot order
    if (++c->stamp == 0u) c->stamp = 1u;
  }
  DFrameParams fp = frame_params(c, sample_limit, first_sample);
  fp.respawn = respawn ? 1u : 0u;
  if (per_slot) timed_launch(c, K_SAMPLE, [&] { launch_sample_ordered(cfg, c->sc, c->st, fp); });
  else timed_launch(c, K_SAMPLE, [&] { launch_sample(cfg, c->sc, c->st, fp); });
  timed_launch(c, K_EXTEND, [&] { launch_extend(cfg, c->sc, c->st, fp); });
  if (c->params.shade_mode == MIRT_SHADE_CLASSIC && !(c->params.nee && fp.have_lights)) {
    // no light sampling: nothing sits between the extend epilogue and the material, one launch does both
    timed_launch(c, K_SHADE, [&] { launch_finish_shade_classic(cfg, c->sc, c->st, fp); });
  } else if (c->params.shade_mode == MIRT_SHADE_CLASSIC) {
    // light sampling: the epilogue with world-space hit records and the light sample, the shadow rays, then
    // the materials — in one kernel, or routed to one kernel per material
    timed_launch(c, K_FINISH, [&] { launch_extend_finish(cfg, c->sc, c->st, fp); });
    timed_launch(c, K_CONNECT, [&] { launch_connect(cfg, c->sc, c->st, fp); });
    if (c->params.stage_kernels != 0u) timed_launch(c, K_SHADE, [&] { launch_shade_classic(cfg, c->sc, c->st, fp); });
    else timed_launch(c, K_SHADE, [&] { launch_connect_shade(cfg, c->sc, c->st, fp); });
  } else {
    // uber shading: the extend epilogue and the shade kernel are one launch
    timed_launch(c, K_SHADE, [&] { launch_finish_shade_uber(cfg, c->sc, c->st, fp); });
  }
  c->frame++;
  c->frames_total++;
}

// Read the device counters (stats then queue counts) into pinned memory.
int fetch_counters(mirt_ctx *c) {
  HIP_TRY(c, hipMemcpyAsync(c->h_stats, c->st.stats, S_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_stats + S_COUNT, c->st.qcount, Q_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  ev_collect(c);
  return MIRT_OK;
}

}  // namespace

extern "C" {

const char *mirt_status_string(int status) {
  switch (status) {
    case MIRT_OK: return "ok";
    case MIRT_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MIRT_ERR_NO_DEVICE: return "no HIP device";
    case MIRT_ERR_OUT_OF_MEMORY: return "out of device memory";
    case MIRT_ERR_HIP: return "HIP runtime error";
    case MIRT_ERR_NOT_READY: return "scene, state or camera not set";
    case MIRT_ERR_BAD_SCENE: return "scene buffers are inconsistent";
    case MIRT_ERR_INTERNAL: return "internal error";
    default: return "unknown status";
  }
}

const char *mirt_last_error(const mirt_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

void mirt_default_params(MirtParams *out) {
  if (!out) return;
  out->max_bounces = 128;          // shaders/sample.slang:71
  out->sky_radiance = 10.0f;       // shaders/ray_extend.slang:36
  out->fixed_point_scale = 1000.0f;// shaders/sample.slang:45-47
  out->exposure = 0.17677669529663687f;  // exp2(-2.5), shaders/sample.slang:54
  out->shade_mode = MIRT_SHADE_UBER;
  out->rng_mode = MIRT_RNG_PER_SAMPLE;
  out->seed = 0x5EED0000u;
  out->nee = 0;
  out->adaptive = 0;
  out->stage_kernels = 0;
}

int mirt_create(int device_ordinal, mirt_ctx **out_ctx) try {
  if (!out_ctx) return MIRT_ERR_INVALID_ARGUMENT;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIRT_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= n) return MIRT_ERR_INVALID_ARGUMENT;
  mirt_ctx *c = new (std::nothrow) mirt_ctx();
  if (!c) return MIRT_ERR_OUT_OF_MEMORY;
  c->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess) { delete c; return MIRT_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MIRT_ERR_HIP; }
  if (hipHostMalloc((void **)&c->h_stats, (S_COUNT + Q_COUNT) * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void **)&c->d_trace_counts, 6 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return MIRT_ERR_OUT_OF_MEMORY;
  }
  c->static_ws = static_scratch(device_ordinal);  // a failure is reported by the entry point that needs it
  mirt_default_params(&c->params);
  *out_ctx = c;
  return MIRT_OK;
} catch (...) {
  return on_exception(nullptr);
}

int mirt_destroy(mirt_ctx *ctx) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_all(ctx->scene_allocs);
  free_all(ctx->instance_allocs);
  free_all(ctx->state_allocs);
  for (EventPair &ep : ctx->ev_free) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  for (EventPair &ep : ctx->ev_pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
  if (ctx->d_trace_counts) (void)hipFree(ctx->d_trace_counts);
  if (ctx->build_ws) (void)hipFree(ctx->build_ws);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

// Geometry: the node array (a reserved TLAS block first, then one BLAS block
// per geometry; every block starts at an odd element, so the child pair
// (l, l+1), l odd, of a node shares a 64-B line) and the triangles, in one
// allocation that the traversal kernels address through one buffer descriptor
// with 32-bit byte offsets; the vertex normals beside it.
static int upload_geometry(mirt_ctx *ctx, const MirtSceneDesc *d, DScene &sc) {
  ctx->geom_node_base.assign(d->n_geometries, 0);
  ctx->geom_tri_base.assign(d->n_geometries, 0);
  ctx->geom_root_start.assign(d->n_geometries, 0);
  ctx->geom_root_count.assign(d->n_geometries, 0);
  ctx->geom_aabbs6.clear();
  // Host staging is written exactly once: the big arrays are left uninitialised
  // (value-initialising 150 MB costs more than packing it) and the triangles are
  // packed straight into the geometry buffer once its layout is known.
  std::vector<DNode> nodes;
  std::unique_ptr<DTriN[]> trins(new DTriN[d->n_indices ? d->n_indices : 1]);
  struct ScanLeaf {
    uint32_t first_tri, count;
    size_t block;
    uint32_t vertex_base;  // of its geometry, for the index rows
  };
  std::vector<ScanLeaf> scan_leaves;
  size_t scan_blocks = 0;
  std::vector<std::pair<uint32_t, uint32_t>> big_sizes;  // first triangle of a leaf with an out-of-line size, the size
  // room for the TLAS to grow without moving the BLAS blocks
  ctx->tlas_capacity = d->n_tlas_nodes + d->n_tlas_nodes / 2u + 16u;
  nodes.reserve((size_t)ctx->tlas_capacity + d->n_blas_nodes + 2 * d->n_geometries + 3);
  nodes.push_back(DNode{});  // element 0 unused: every block starts at an odd element
  sc.tlas_base = (uint32_t)nodes.size();
  nodes.resize(nodes.size() + ctx->tlas_capacity);
  for (uint32_t g = 0; g < d->n_geometries; ++g) {
    const MirtGeometryOffsets &o = d->geometry_offsets[g];
    const bool last = g + 1 == d->n_geometries;
    const uint32_t node_end = last ? d->n_blas_nodes : d->geometry_offsets[g + 1].blas_node;
    const uint32_t nn = node_end - o.blas_node;
    if ((nodes.size() & 1u) == 0u) nodes.push_back(DNode{});
    ctx->geom_node_base[g] = (uint32_t)nodes.size();
    ctx->geom_tri_base[g] = o.index;
    // light sampling draws from the primitive range of the BLAS root
    // (archive/light.slang:60-64).  start/end of an inner node mean nothing
    // (src/bvh.rs:206) and are not validated, so an inner root stands for every
    // triangle of its geometry — which is what the reference's builder leaves
    // in them (src/bvh.rs:144-151); a leaf root's range has been validated.
    const MirtBVHNode &root = d->blas_nodes[o.blas_node];
    ctx->geom_aabbs6.insert(ctx->geom_aabbs6.end(), root.lb, root.lb + 3);
    ctx->geom_aabbs6.insert(ctx->geom_aabbs6.end(), root.ub, root.ub + 3);
    const uint32_t idx_end_g = last ? d->n_indices : d->geometry_offsets[g + 1].index;
    ctx->geom_root_start[g] = root.is_leaf == 1 ? root.start : 0u;
    ctx->geom_root_count[g] = root.is_leaf == 1 ? root.end - root.start : idx_end_g - o.index;
    const size_t at = nodes.size();
    nodes.resize(at + nn);
    std::vector<uint32_t> sizes(nn);
    pack_nodes(d->blas_nodes + o.blas_node, nn, nodes.data() + at, sizes.data(), false);
    // leaves the whole wave scans get a second, lane-interleaved copy of their triangles;
    // out-of-line leaf sizes ride in the first triangle of the leaf (written below)
    for (uint32_t i = 0; i < nn; ++i) {
      const DNode &dn = nodes[at + i];
      const uint32_t tag = dn.b >> 24;
      if (tag == kBigLeaf && sizes[i] > 0) big_sizes.push_back({o.index + dn.a, sizes[i]});
      if (tag >= kScanLeafMin) {
        const uint32_t cnt = tag == kBigLeaf ? sizes[i] : tag;
        scan_leaves.push_back(ScanLeaf{o.index + dn.a, cnt, scan_blocks, o.vertex});
        scan_blocks += (cnt + 63u) / 64u;
      }
    }
  }
  if ((nodes.size() & 1u) == 0u) nodes.push_back(DNode{});
  const size_t node_bytes = nodes.size() * sizeof(DNode);
  const size_t tri_off = (node_bytes + 255u) & ~(size_t)255u;
  const size_t scan_off = (tri_off + (size_t)d->n_indices * sizeof(DTri) + 64u + 255u) & ~(size_t)255u;
  // a second copy of the triangles without the padding words, for the fetches single lanes make (three quarters of
  // the bytes, and a leaf of four reaches over three 64-byte sectors at most instead of four)
  const size_t tri36_off = (scan_off + scan_blocks * (size_t)kScanBlockBytes + 64u + 255u) & ~(size_t)255u;
  const size_t geo_bytes = tri36_off + (size_t)d->n_indices * 36u + 64u;  // tail pad: reads past the last record stay inside
  if (geo_bytes >= ((size_t)1 << 32))
    return fail(ctx, MIRT_ERR_BAD_SCENE, "nodes and triangles exceed the 4 GiB the traversal kernels address");
  std::unique_ptr<unsigned char[]> geo(new unsigned char[geo_bytes]);
  memcpy(geo.get(), nodes.data(), node_bytes);
  memset(geo.get() + node_bytes, 0, tri_off - node_bytes);
  const size_t tri_end = tri_off + (size_t)d->n_indices * sizeof(DTri);
  memset(geo.get() + tri_end, 0, geo_bytes - tri_end);  // pads and the scan blocks' unused lanes
  DTri *tris = reinterpret_cast<DTri *>(geo.get() + tri_off);
  for (uint32_t g = 0; g < d->n_geometries; ++g) {
    const MirtGeometryOffsets &o = d->geometry_offsets[g];
    const uint32_t idx_end = g + 1 == d->n_geometries ? d->n_indices : d->geometry_offsets[g + 1].index;
    for (uint32_t p = o.index; p < idx_end; ++p) {
      const MirtIndex &f = d->indices[p];
      const MirtVertex &v0 = d->vertices[o.vertex + f.v[0]];
      const MirtVertex &v1 = d->vertices[o.vertex + f.v[1]];
      const MirtVertex &v2 = d->vertices[o.vertex + f.v[2]];
      DTri t;
      memset(&t, 0, sizeof(t));
      t.p0x = v0.position[0]; t.p0y = v0.position[1]; t.p0z = v0.position[2];
      t.e1x = v1.position[0] - v0.position[0]; t.e1y = v1.position[1] - v0.position[1]; t.e1z = v1.position[2] - v0.position[2];
      t.e2x = v2.position[0] - v0.position[0]; t.e2y = v2.position[1] - v0.position[1]; t.e2z = v2.position[2] - v0.position[2];
      tris[p] = t;
      const float packed[9] = {t.p0x, t.p0y, t.p0z, t.e1x, t.e1y, t.e1z, t.e2x, t.e2y, t.e2z};
      memcpy(geo.get() + tri36_off + (size_t)p * 36u, packed, 36);
      DTriN n;
      memcpy(n.n0, v0.normal, 16); memcpy(n.n1, v1.normal, 16); memcpy(n.n2, v2.normal, 16);
      trins[p] = n;
    }
  }
  for (const auto &bs : big_sizes) tris[bs.first].p0w = bs.second;
  // Scan blocks: 64 triangles each, as three planes a wave reads with one
  // contiguous load each — plane 0 (p0.xyz, e1.x) and plane 1 (e1.yz, e2.xy)
  // 16 bytes per lane, plane 2 (e2.z) 4 bytes per lane.  The byte offset of a
  // leaf's first block rides in the first triangle of the leaf.
  for (const ScanLeaf &sl : scan_leaves) {
    const size_t first = scan_off + sl.block * (size_t)kScanBlockBytes;
    tris[sl.first_tri].e1w = (uint32_t)first;
    for (uint32_t k = 0; k < sl.count; ++k) {
      const DTri &t = tris[sl.first_tri + k];
      unsigned char *blk = geo.get() + first + (k / 64u) * (size_t)kScanBlockBytes;
      const uint32_t lane = k % 64u;
      const float plane0[4] = {t.p0x, t.p0y, t.p0z, t.e1x}, plane1[4] = {t.e1y, t.e1z, t.e2x, t.e2y};
      memcpy(blk + lane * 16u, plane0, 16);
      memcpy(blk + 1024u + lane * 16u, plane1, 16);
      memcpy(blk + 2048u + lane * 4u, &t.e2z, 4);
    }
  }
  // LDS leaf cache: the biggest scanned leaves first, as long as they fit the image —
  // direct form: a copy of their scan blocks;
  std::vector<uint4> cache_image;
  if (kLeafCacheDirect) {
    std::vector<const ScanLeaf *> order;
    for (const ScanLeaf &sl : scan_leaves) order.push_back(&sl);
    std::stable_sort(order.begin(), order.end(), [](const ScanLeaf *a, const ScanLeaf *b) { return a->count > b->count; });
    size_t at = 0;
    std::vector<std::pair<const ScanLeaf *, size_t>> taken;
    for (const ScanLeaf *sl : order) {
      const size_t bytes = (size_t)((sl->count + 63u) / 64u) * kScanBlockBytes;
      if (at + bytes > kLeafCacheBytes) continue;
      taken.push_back({sl, at});
      at += bytes;
    }
    cache_image.assign(at / 16u, uint4{0u, 0u, 0u, 0u});
    static_assert(kScanBlockBytes % 16u == 0u, "scan blocks are whole 16-byte words");
    for (const auto &tk : taken) {
      const ScanLeaf *sl = tk.first;
      const size_t bytes = (size_t)((sl->count + 63u) / 64u) * kScanBlockBytes;
      memcpy(reinterpret_cast<unsigned char *>(cache_image.data()) + tk.second,
             geo.get() + scan_off + sl->block * (size_t)kScanBlockBytes, bytes);
      tris[sl->first_tri].e2w = (uint32_t)tk.second + 1u;
    }
  } else {
    // indexed form: their unique vertices and index rows
    std::vector<const ScanLeaf *> order;
    for (const ScanLeaf &sl : scan_leaves) order.push_back(&sl);
    std::stable_sort(order.begin(), order.end(), [](const ScanLeaf *a, const ScanLeaf *b) { return a->count > b->count; });
    std::unordered_map<uint32_t, uint32_t> slot_of;  // global vertex index -> vertex number in the image
    std::vector<uint32_t> image_vertices;            // global vertex index of every image vertex
    std::vector<const ScanLeaf *> taken;
    size_t taken_tris = 0;
    for (const ScanLeaf *sl : order) {
      if ((taken_tris + sl->count) * 8u > kLeafCacheBytes) continue;
      std::vector<uint32_t> fresh;  // vertices this leaf would add, in order of first use
      std::unordered_set<uint32_t> fresh_set;
      const size_t max_vertices = kLeafCacheBytes / 16u;
      for (uint32_t k = 0; k < sl->count && image_vertices.size() + fresh.size() <= max_vertices; ++k)
        for (int j = 0; j < 3; ++j) {
          const uint32_t v = sl->vertex_base + d->indices[sl->first_tri + k].v[j];
          if (!slot_of.count(v) && fresh_set.insert(v).second) fresh.push_back(v);
        }
      const size_t nv = image_vertices.size() + fresh.size();
      if (nv > 65536u || nv * 16u + (taken_tris + sl->count) * 8u > kLeafCacheBytes) continue;
      for (uint32_t v : fresh) {
        slot_of[v] = (uint32_t)image_vertices.size();
        image_vertices.push_back(v);
      }
      taken.push_back(sl);
      taken_tris += sl->count;
    }
    if (!taken.empty()) {
      const size_t vert_bytes = image_vertices.size() * 16u;
      cache_image.assign((vert_bytes + taken_tris * 8u + 15u) / 16u, uint4{0u, 0u, 0u, 0u});
      unsigned char *img = reinterpret_cast<unsigned char *>(cache_image.data());
      for (size_t i = 0; i < image_vertices.size(); ++i) memcpy(img + i * 16u, d->vertices[image_vertices[i]].position, 12);
      size_t at = vert_bytes;
      for (const ScanLeaf *sl : taken) {
        tris[sl->first_tri].e2w = (uint32_t)at + 1u;
        for (uint32_t k = 0; k < sl->count; ++k) {
          const MirtIndex &f = d->indices[sl->first_tri + k];
          const uint32_t v0 = slot_of[sl->vertex_base + f.v[0]], v1 = slot_of[sl->vertex_base + f.v[1]],
                         v2 = slot_of[sl->vertex_base + f.v[2]];
          const uint32_t row[2] = {v0 | (v1 << 16), v2};
          memcpy(img + at, row, 8);
          at += 8u;
        }
      }
    }
  }
  unsigned char *pg = nullptr;
  int rc = dev_alloc(ctx, ctx->scene_allocs, &pg, geo_bytes);
  if (rc != MIRT_OK) return rc;
  HIP_TRY(ctx, copy_sync(ctx->stream, pg, geo.get(), geo_bytes, hipMemcpyHostToDevice));
  sc.geo = pg;
  sc.geo_bytes = (uint32_t)geo_bytes;
  sc.tri_off = (uint32_t)tri_off;
  sc.tri36_off = (uint32_t)tri36_off;
  sc.nodes = reinterpret_cast<const DNode *>(pg);
  sc.tri = reinterpret_cast<const DTri *>(pg + tri_off);
  sc.leaf_cache = nullptr;
  sc.leaf_cache_bytes = (uint32_t)(cache_image.size() * 16u);
  if (!cache_image.empty() && (rc = dev_upload(ctx, ctx->scene_allocs, &sc.leaf_cache, cache_image)) != MIRT_OK) return rc;
  DTriN *ptn = nullptr;
  if ((rc = dev_alloc(ctx, ctx->scene_allocs, &ptn, d->n_indices)) != MIRT_OK) return rc;
  if (d->n_indices) HIP_TRY(ctx, copy_sync(ctx->stream, ptn, trins.get(), (size_t)d->n_indices * sizeof(DTriN), hipMemcpyHostToDevice));
  sc.trin = ptn;
  sc.tri_uv = nullptr;
  if (d->n_textures) {
    // the three vertex uvs of every triangle, two float4 per triangle
    std::unique_ptr<float[]> uvs(new float[(size_t)8 * (d->n_indices ? d->n_indices : 1)]);
    for (uint32_t g = 0; g < d->n_geometries; ++g) {
      const MirtGeometryOffsets &o = d->geometry_offsets[g];
      const uint32_t idx_end = g + 1 == d->n_geometries ? d->n_indices : d->geometry_offsets[g + 1].index;
      for (uint32_t p = o.index; p < idx_end; ++p) {
        float *q = uvs.get() + (size_t)8 * p;
        for (int k = 0; k < 3; ++k) {
          const MirtVertex &v = d->vertices[o.vertex + d->indices[p].v[k]];
          q[2 * k] = v.uv[0];
          q[2 * k + 1] = v.uv[1];
        }
        q[6] = q[7] = 0.0f;
      }
    }
    float *puv = nullptr;
    if ((rc = dev_alloc(ctx, ctx->scene_allocs, &puv, (size_t)8 * d->n_indices)) != MIRT_OK) return rc;
    if (d->n_indices) HIP_TRY(ctx, copy_sync(ctx->stream, puv, uvs.get(), (size_t)32 * d->n_indices, hipMemcpyHostToDevice));
    sc.tri_uv = reinterpret_cast<const float4 *>(puv);
  }
  return MIRT_OK;
}

// Everything that depends on instances, transforms or materials: the TLAS
// block of the node array (written in place), the TLAS-leaf slot list, the
// instance records with their matrices, materials, lights, per-material tables.
static int upload_instances(mirt_ctx *ctx, const MirtSceneDesc *d, DScene &sc) {
  auto &own = ctx->instance_allocs;
  int rc;
  {
    std::vector<DNode> tlas(d->n_tlas_nodes);
    std::vector<uint32_t> tlas_sizes(d->n_tlas_nodes);
    pack_nodes(d->tlas_nodes, d->n_tlas_nodes, tlas.data(), tlas_sizes.data(), true);
    HIP_TRY(ctx, copy_sync(ctx->stream, const_cast<DNode *>(sc.nodes) + sc.tlas_base, tlas.data(), tlas.size() * sizeof(DNode),
                           hipMemcpyHostToDevice));
    // a leaf of kBigLeaf or more slots keeps its size out of line, indexed by its first slot
    // both slot tables have room for one slot per instance even when the caller's TLAS lists fewer: a tree rebuilt
    // on the device (mirt_scene_update_transforms) lists every instance
    const size_t slot_room = std::max<size_t>(std::max(d->n_tlas_to_instances, d->n_instances), 1u);
    std::vector<uint32_t> big(slot_room, 0u);
    for (uint32_t i = 0; i < d->n_tlas_nodes; ++i)
      if ((tlas[i].b >> 24) == kBigLeaf) big[tlas[i].a] = tlas_sizes[i];
    if ((rc = dev_upload(ctx, own, &sc.tlas_big, big)) != MIRT_OK) return rc;
    std::vector<uint32_t> t2i(d->tlas_to_instances, d->tlas_to_instances + d->n_tlas_to_instances);
    t2i.resize(slot_room, 0u);
    if ((rc = dev_upload(ctx, own, &sc.tlas_to_inst, t2i)) != MIRT_OK) return rc;
    ctx->n_tlas_slots = d->n_tlas_to_instances;
  }
  {
    std::vector<DInst> insts(d->n_instances);
    parallel_for(d->n_instances, [&](uint32_t i) {
      const MirtInstance &in = d->instances[i];
      DInst di;
      memset(&di, 0, sizeof(di));
      instance_matrices(d->transforms[in.transform], di.m, di.mi);
      di.node_base = ctx->geom_node_base[in.geometry];
      di.tri_base = ctx->geom_tri_base[in.geometry];
      di.material = in.material;
      if (d->n_classic_refs) {
        di.kind = d->classic_refs[in.material].kind;
        di.kind_data = d->classic_refs[in.material].data;
      }
      di.root_start = ctx->geom_root_start[in.geometry];
      di.root_count = ctx->geom_root_count[in.geometry];
      insts[i] = di;
    });
    if ((rc = dev_upload(ctx, own, &sc.inst, insts)) != MIRT_OK) return rc;
    ctx->host_insts = insts;
    ctx->host_instances.assign(d->instances, d->instances + d->n_instances);
    ctx->n_transforms = d->n_transforms;
  }
  {
    std::vector<DMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
      DMaterial m;
      memset(&m, 0, sizeof(m));
      memcpy(m.colour, d->materials[i].colour, 16);
      memcpy(m.emissive, d->materials[i].emissive, 16);
      m.metallic = d->materials[i].metallic;
      m.roughness = d->materials[i].roughness;
      m.ior = d->materials[i].ior;
      m.transmission = d->materials[i].transmission;
      m.tex[0] = d->materials[i].colour_texture;
      m.tex[1] = d->materials[i].emissive_texture;
      mats[i] = m;
    }
    {
      std::vector<uint4> tex(d->n_textures);
      for (uint32_t i = 0; i < d->n_textures; ++i)
        tex[i] = uint4{d->textures[i].width, d->textures[i].height, d->textures[i].first_texel, d->textures[i].filter};
      std::vector<uint32_t> texels;
      if (d->n_textures) texels.assign(d->texels, d->texels + d->n_texels);
      if ((rc = dev_upload(ctx, own, &sc.textures, tex)) != MIRT_OK) return rc;
      if ((rc = dev_upload(ctx, own, &sc.texels, texels)) != MIRT_OK) return rc;
      sc.n_textures = d->n_textures;
    }
    if ((rc = dev_upload(ctx, own, &sc.mat, mats)) != MIRT_OK) return rc;
    const bool no_lights = d->n_light_sources == 0 || (d->n_light_sources == 1 && d->light_sources[0] == 0xFFFFFFFFu);
    std::vector<uint32_t> lights;
    if (!no_lights) lights.assign(d->light_sources, d->light_sources + d->n_light_sources);
    sc.n_lights = (uint32_t)lights.size();
    if ((rc = dev_upload(ctx, own, &sc.lights, lights)) != MIRT_OK) return rc;
    std::vector<float> lam, met, die, emi;
    if (d->n_classic_refs) {
      if (d->lambertian) lam.assign((const float *)d->lambertian, (const float *)d->lambertian + 4 * (size_t)d->n_lambertian);
      if (d->metallic) met.assign((const float *)d->metallic, (const float *)d->metallic + 8 * (size_t)d->n_metallic);
      if (d->dielectric) die.assign((const float *)d->dielectric, (const float *)d->dielectric + 8 * (size_t)d->n_dielectric);
      if (d->emissive) emi.assign((const float *)d->emissive, (const float *)d->emissive + 4 * (size_t)d->n_emissive);
    }
    if ((rc = dev_upload(ctx, own, &sc.lambertian, lam)) != MIRT_OK) return rc;
    if ((rc = dev_upload(ctx, own, &sc.metallic, met)) != MIRT_OK) return rc;
    if ((rc = dev_upload(ctx, own, &sc.dielectric, die)) != MIRT_OK) return rc;
    if ((rc = dev_upload(ctx, own, &sc.emissive, emi)) != MIRT_OK) return rc;
  }
  return MIRT_OK;
}

int mirt_scene_upload(mirt_ctx *ctx, const MirtSceneDesc *d) try {
  ENTER(ctx);
  if (!d) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "scene is null");
  std::string why;
  if (!validate_scene(d, why)) return fail(ctx, MIRT_ERR_BAD_SCENE, why);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_all(ctx->scene_allocs);
  free_all(ctx->instance_allocs);
  ctx->have_scene = false;
  DScene sc{};
  int rc = upload_geometry(ctx, d, sc);
  if (rc == MIRT_OK) rc = upload_instances(ctx, d, sc);
  if (rc != MIRT_OK) return rc;
  ctx->sc = sc;
  ctx->n_instances = d->n_instances;
  ctx->n_geometries = d->n_geometries;
  ctx->have_classic_tables = d->n_classic_refs != 0;
  ctx->have_scene = true;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_scene_update_instances(mirt_ctx *ctx, const MirtSceneDesc *d) try {
  ENTER(ctx);
  if (!d) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "scene is null");
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  // a different set of geometries, or a TLAS that outgrew its block: start over
  // (so does a first texture: the per-triangle uvs are built from the geometry arrays, which must then be valid)
  if (d->n_geometries != ctx->n_geometries || d->n_tlas_nodes > ctx->tlas_capacity ||
      (d->n_textures != 0 && ctx->sc.tri_uv == nullptr))
    return mirt_scene_upload(ctx, d);
  std::string why;
  if (!validate_instances(d, why)) return fail(ctx, MIRT_ERR_BAD_SCENE, why);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_all(ctx->instance_allocs);
  ctx->have_scene = false;
  DScene sc = ctx->sc;
  const int rc = upload_instances(ctx, d, sc);
  if (rc != MIRT_OK) return rc;
  ctx->sc = sc;
  ctx->n_instances = d->n_instances;
  ctx->have_classic_tables = d->n_classic_refs != 0;
  ctx->have_scene = true;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_scene_update_transforms(mirt_ctx *ctx, const MirtTransform *transforms, uint32_t n_transforms, float *device_ms) try {
  ENTER(ctx);
  if (device_ms) *device_ms = 0.0f;
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  if (!transforms || n_transforms != ctx->n_transforms)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "update_transforms: as many transforms as the scene was uploaded with");
  const uint32_t n = ctx->n_instances;
  // The rebuilt tree lists every instance once.  A scene whose uploaded TLAS lists another number of slots (the
  // caller left instances out, or listed one twice) is not this call's to rebuild: it would change which instances
  // are visible.
  if (ctx->n_tlas_slots != n)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT,
                "update_transforms: the uploaded TLAS lists " + std::to_string(ctx->n_tlas_slots) + " slots for " +
                    std::to_string(n) + " instances; use mirt_scene_update_instances");
  // instance matrices on the host (its sin / cos), as in a full upload
  std::vector<DInst> insts = ctx->host_insts;
  std::vector<float> mats((size_t)16 * n);
  std::vector<uint32_t> geom(n);
  parallel_for(n, [&](uint32_t i) {
    DInst &di = insts[i];
    instance_matrices(transforms[ctx->host_instances[i].transform], di.m, di.mi);
    float *c = &mats[(size_t)16 * i];
    for (int col = 0; col < 4; ++col)
      for (int r = 0; r < 3; ++r) c[4 * col + r] = di.m[4 * r + col];
    c[3] = c[7] = c[11] = 0.0f;
    c[15] = 1.0f;
    geom[i] = ctx->host_instances[i].geometry;
  });
  // boxes, tree and the traversal-layout TLAS on the device, written over the old one in place (stream order
  // keeps frames already submitted on the old TLAS); nothing of the scene is touched when the new tree does
  // not fit the room reserved for it
  TlasResident res;
  res.nodes = const_cast<DNode *>(ctx->sc.nodes) + ctx->sc.tlas_base;
  res.capacity = ctx->tlas_capacity;
  res.big = const_cast<uint32_t *>(ctx->sc.tlas_big);
  res.to_inst = const_cast<uint32_t *>(ctx->sc.tlas_to_inst);
  uint32_t n_nodes = 0;
  std::string err;
  const int rc = build_tlas_device(ctx->stream, ctx->geom_aabbs6.data(), ctx->n_geometries, mats.data(), geom.data(), n,
                                   nullptr, 0, &n_nodes, nullptr, device_ms, &ctx->build_ws, &ctx->build_ws_bytes, err, &res);
  if (rc != MIRT_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    return fail(ctx, rc, err);
  }
  // the instance records with their new matrices, in one copy behind the tree
  HIP_TRY(ctx, hipMemcpyAsync(const_cast<DInst *>(ctx->sc.inst), insts.data(), (size_t)n * sizeof(DInst),
                              hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the host buffers of the copies above must outlive them
  ctx->host_insts.swap(insts);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_state_create(mirt_ctx *ctx, uint32_t width, uint32_t height, uint32_t threads,
                      const MirtSampleSource *sources, uint32_t n_sources, const MirtParams *params) try {
  ENTER(ctx);
  if (!width || !height || !threads || !sources || !n_sources || !params)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_create: zero size or null pointer");
  if (params->max_bounces == 0 || params->shade_mode > MIRT_SHADE_CLASSIC || params->rng_mode > MIRT_RNG_PER_SLOT ||
      params->adaptive > 1u)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_create: bad params");
  if (!(params->fixed_point_scale > 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_create: fixed_point_scale must be > 0");
  for (uint32_t i = 0; i < n_sources; ++i)
    if (sources[i].out_pos[0] >= width || sources[i].out_pos[1] >= height)
      return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_create: sample source outside the image");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_all(ctx->state_allocs);
  ctx->have_state = false;
  DState st{};
  st.slots = threads;
  st.n_sources = n_sources;
  st.width = width;
  st.height = height;
  int rc;
  auto &own = ctx->state_allocs;
  if ((rc = dev_alloc(ctx, own, &st.ray, threads)) != MIRT_OK) return rc;

  if ((rc = dev_alloc(ctx, own, &st.path, threads)) != MIRT_OK) return rc;
  // what only some shading paths touch is only allocated for them (84 of a slot's 180 bytes): the connect
  // stage's records (world-space hit, ShadowData) and queue with per-material shading and light sampling, the
  // four material queues with one kernel per material on top of that
  const bool connect_stage = params->shade_mode == MIRT_SHADE_CLASSIC && params->nee != 0u;
  const bool material_queues = connect_stage && params->stage_kernels != 0u;
  if ((rc = dev_alloc(ctx, own, &st.hit, connect_stage ? threads : 0u)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &st.shadow, connect_stage ? threads : 0u)) != MIRT_OK) return rc;
  for (int q = 0; q < Q_COUNT; ++q) {
    const bool used = q < Q_CONNECT || (q == Q_CONNECT ? connect_stage : material_queues);
    if ((rc = dev_alloc(ctx, own, &st.qdata[q], used ? threads : 0u)) != MIRT_OK) return rc;
  }
  if ((rc = dev_alloc(ctx, own, &st.qcount, QW_COUNT)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &st.stats, S_COUNT)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &st.sample_index, 1)) != MIRT_OK) return rc;
  MirtSampleSource *dsrc = nullptr;
  if ((rc = dev_alloc(ctx, own, &dsrc, n_sources)) != MIRT_OK) return rc;
  HIP_TRY(ctx, copy_sync(ctx->stream, dsrc, sources, (size_t)n_sources * sizeof(MirtSampleSource), hipMemcpyHostToDevice));
  st.sources = dsrc;
  if ((rc = dev_alloc(ctx, own, &st.sum3, (size_t)kSumStride * n_sources)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &ctx->d_export4, (size_t)4 * n_sources)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &st.count, n_sources)) != MIRT_OK) return rc;
  if ((rc = dev_alloc(ctx, own, &st.output, (size_t)width * height)) != MIRT_OK) return rc;
  HIP_TRY(ctx, hipMemsetAsync(st.sum3, 0, (size_t)8 * kSumStride * n_sources, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(st.count, 0, (size_t)4 * n_sources, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(st.output, 0, (size_t)4 * width * height, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(st.stats, 0, S_COUNT * sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(st.qcount, 0, QW_COUNT * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(st.sample_index, 0, sizeof(unsigned long long), ctx->stream));
  st.n_active = n_sources;
  st.active = nullptr;
  ctx->d_active = nullptr;
  if ((rc = dev_alloc(ctx, own, &st.flags, n_sources)) != MIRT_OK) return rc;
  HIP_TRY(ctx, hipMemsetAsync(st.flags, 0, (size_t)4 * n_sources, ctx->stream));
  if (params->adaptive) {
    if ((rc = dev_alloc(ctx, own, &st.moments, (size_t)2 * n_sources)) != MIRT_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(st.moments, 0, (size_t)16 * n_sources, ctx->stream));
    if ((rc = dev_alloc(ctx, own, &ctx->d_active, n_sources)) != MIRT_OK) return rc;
    // sources the caller hands over flagged are skipped from the start
    std::vector<uint32_t> flags(n_sources), act;
    bool any = false;
    for (uint32_t i = 0; i < n_sources; ++i) {
      flags[i] = sources[i].flags & 1u;
      any = any || flags[i] != 0u;
    }
    if (any) {
      for (uint32_t i = 0; i < n_sources; ++i)
        if (!flags[i]) act.push_back(i);
      HIP_TRY(ctx, copy_sync(ctx->stream, st.flags, flags.data(), (size_t)4 * n_sources, hipMemcpyHostToDevice));
      if (!act.empty()) HIP_TRY(ctx, copy_sync(ctx->stream, ctx->d_active, act.data(), act.size() * 4, hipMemcpyHostToDevice));
      st.active = ctx->d_active;
      st.n_active = (uint32_t)act.size();
    }
  }
  if (params->rng_mode == MIRT_RNG_PER_SLOT) {
    if ((rc = dev_alloc(ctx, own, &st.slot_stamp, threads)) != MIRT_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(st.slot_stamp, 0, (size_t)4 * threads, ctx->stream));
    const size_t ws_words = 8u + ((size_t)threads + kStreamBlock - 1u) / kStreamBlock;
    if ((rc = dev_alloc(ctx, own, &st.ordered_ws, ws_words)) != MIRT_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(st.ordered_ws, 0, 4 * ws_words, ctx->stream));
  }
  ctx->st = st;
  ctx->params = *params;
  ctx->have_state = true;
  ctx->next_sample = 0;
  ctx->frame = 0;
  ctx->slots_seeded = false;
  ctx->rng_seeded = false;
  ctx->stamp = 0;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_state_set_rng(mirt_ctx *ctx, const uint32_t (*rng)[4], uint32_t n_slots) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (ctx->params.rng_mode != MIRT_RNG_PER_SLOT)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_set_rng: the state was created with per-sample streams");
  if (!rng || n_slots != ctx->st.slots) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "state_set_rng: one uint4 per path slot");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the stream state is the third 16-byte quarter of a slot's path record: the third plane, or bytes 32..47 of
  // every 48-byte record
  if (kPathPlanes)
    HIP_TRY(ctx, hipMemcpy(reinterpret_cast<unsigned char *>(ctx->st.path) + (size_t)32 * n_slots, rng, (size_t)16 * n_slots,
                           hipMemcpyHostToDevice));
  else
    HIP_TRY(ctx, hipMemcpy2D(reinterpret_cast<unsigned char *>(ctx->st.path) + 32, sizeof(DPath), rng, 16, 16, n_slots,
                             hipMemcpyHostToDevice));
  ctx->rng_seeded = true;
  ctx->slots_seeded = false;  // in-flight paths are dropped
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

// The host half of a convergence pass: with `flags` on the device (set there by k_adapt_update, or uploaded here
// when `upload` is given: the flags another member of a group decided), the unflagged sources become the
// ascending active list that spawn_sample draws from.
static int install_flags(mirt_ctx *ctx, const uint32_t *upload, std::vector<uint32_t> *flags_out, uint32_t *n_active) {
  // Sample numbers already handed out must not come back under the new list:
  // continue behind the highest one used so far.
  unsigned long long handed = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&handed, ctx->st.sample_index, sizeof(handed), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<uint32_t> flags(ctx->st.n_sources), act;
  if (upload) {
    flags.assign(upload, upload + ctx->st.n_sources);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st.flags, flags.data(), (size_t)4 * ctx->st.n_sources, hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(flags.data(), ctx->st.flags, (size_t)4 * ctx->st.n_sources, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ev_collect(ctx);
  if (ctx->slots_seeded && ctx->st.n_active != 0u) {  // mirt_step frames in flight: their numbering restarts
    ctx->next_sample += (uint32_t)((handed + ctx->st.n_active - 1ull) / ctx->st.n_active);
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.sample_index, 0, sizeof(unsigned long long), ctx->stream));
  }
  act.reserve(ctx->st.n_sources);
  for (uint32_t i = 0; i < ctx->st.n_sources; ++i)
    if ((flags[i] & 1u) == 0u) act.push_back(i);
  if (act.size() == ctx->st.n_sources) {
    ctx->st.active = nullptr;
  } else {
    if (!act.empty()) HIP_TRY(ctx, copy_sync(ctx->stream, ctx->d_active, act.data(), act.size() * 4, hipMemcpyHostToDevice));
    ctx->st.active = ctx->d_active;
  }
  ctx->st.n_active = (uint32_t)act.size();
  if (n_active) *n_active = ctx->st.n_active;
  if (flags_out) flags_out->swap(flags);
  return MIRT_OK;
}

static int adapt_update_impl(mirt_ctx *ctx, float rel_threshold, uint32_t min_samples, uint32_t *n_active,
                             std::vector<uint32_t> *flags_out) {
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (!ctx->params.adaptive) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "adapt_update: the state was created with adaptive = 0");
  if (!(rel_threshold >= 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "adapt_update: rel_threshold must be >= 0");
  const double thr2 = (double)rel_threshold * (double)rel_threshold;
  launch_adapt_update(launch_cfg(ctx), ctx->st, thr2, min_samples);
  HIP_TRY(ctx, hipGetLastError());
  return install_flags(ctx, nullptr, flags_out, n_active);
}

int mirt_adapt_update(mirt_ctx *ctx, float rel_threshold, uint32_t min_samples, uint32_t *n_active) try {
  ENTER(ctx);
  return adapt_update_impl(ctx, rel_threshold, min_samples, n_active, nullptr);
} catch (...) {
  return on_exception(ctx);
}

int mirt_read_flags(mirt_ctx *ctx, uint32_t *flags, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (!flags || n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "read_flags: n_sources mismatch");
  HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->st.flags, n_sources * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_read_moments(mirt_ctx *ctx, uint64_t *moments, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state || !ctx->st.moments) return fail(ctx, MIRT_ERR_NOT_READY, "no state with adaptive = 1");
  if (!moments || n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "read_moments: n_sources mismatch");
  HIP_TRY(ctx, hipMemcpyAsync(moments, ctx->st.moments, n_sources * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_write_moments(mirt_ctx *ctx, const uint64_t *moments, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state || !ctx->st.moments) return fail(ctx, MIRT_ERR_NOT_READY, "no state with adaptive = 1");
  if (!moments || n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "write_moments: n_sources mismatch");
  HIP_TRY(ctx, hipMemcpyAsync(ctx->st.moments, moments, n_sources * 16, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_camera_set(mirt_ctx *ctx, const MirtCamera *camera) try {
  ENTER(ctx);
  if (!camera) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "camera is null");
  ctx->camera = *camera;
  ctx->have_camera = true;
  if (camera->changed != 0 && ctx->have_state) return mirt_reset_accum(ctx);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_reset_accum(mirt_ctx *ctx) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  // sampleCleanup (shaders/sample.slang:123-145) for every source, including
  // the tail its integer work split leaves out.  Counts restart at 0, not 1.
  HIP_TRY(ctx, hipMemsetAsync(ctx->st.sum3, 0, (size_t)8 * kSumStride * ctx->st.n_sources, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->st.count, 0, (size_t)4 * ctx->st.n_sources, ctx->stream));
  // ... and every source takes samples again (sample.slang:137 clears the flags)
  HIP_TRY(ctx, hipMemsetAsync(ctx->st.flags, 0, (size_t)4 * ctx->st.n_sources, ctx->stream));
  if (ctx->st.moments) HIP_TRY(ctx, hipMemsetAsync(ctx->st.moments, 0, (size_t)16 * ctx->st.n_sources, ctx->stream));
  ctx->st.active = nullptr;
  ctx->st.n_active = ctx->st.n_sources;
  ctx->next_sample = 0;
  ctx->slots_seeded = false;  // in-flight paths are dropped
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_set_sample_range(mirt_ctx *ctx, uint32_t first_sample) try {
  ENTER(ctx);
  ctx->next_sample = first_sample;
  ctx->slots_seeded = false;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

static int check_ready(mirt_ctx *ctx) {
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state created");
  if (!ctx->have_camera) return fail(ctx, MIRT_ERR_NOT_READY, "no camera set");
  if (ctx->params.shade_mode == MIRT_SHADE_CLASSIC && !ctx->have_classic_tables)
    return fail(ctx, MIRT_ERR_NOT_READY, "classic shading needs the per-material tables");
  if (ctx->params.rng_mode == MIRT_RNG_PER_SLOT && !ctx->rng_seeded)
    return fail(ctx, MIRT_ERR_NOT_READY, "per-slot random streams need mirt_state_set_rng");
  return MIRT_OK;
}

int mirt_step(mirt_ctx *ctx, uint32_t n_frames) try {
  ENTER(ctx);
  int rc = check_ready(ctx);
  if (rc != MIRT_OK) return rc;
  if (ctx->st.n_active == 0u) return MIRT_OK;  // every source has converged: nothing to trace
  if (!ctx->slots_seeded && (rc = seed_slots(ctx)) != MIRT_OK) return rc;
  // samples keep regenerating: no limit on the number handed out
  for (uint32_t f = 0; f < n_frames; ++f) run_frame(ctx, ~0ull, ctx->next_sample, false);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_render_spp(mirt_ctx *ctx, uint32_t spp) try {
  ENTER(ctx);
  int rc = check_ready(ctx);
  if (rc != MIRT_OK) return rc;
  if (spp == 0 || ctx->st.n_active == 0u) return MIRT_OK;
  const unsigned long long limit = (unsigned long long)ctx->st.n_active * spp;  // spp for every source that still takes samples
  const uint32_t first = ctx->next_sample;
  // paths left by mirt_step are dropped: every slot starts on a sample of this call
  ctx->frame = 0;
  ctx->slots_seeded = true;
  launch_spawn_first(launch_cfg(ctx), ctx->st, frame_params(ctx, limit, first));
  if ((rc = fetch_counters(ctx)) != MIRT_OK) return rc;
  const unsigned long long finished0 = ctx->h_stats[S_FINISHED];
  // Each frame advances every path by one bounce.  Poll the finished-sample
  // counter every few frames; the loop ends when all samples of this call
  // have been accumulated.
  const uint32_t poll = 4;
  // A sample lives at most max_bounces + 1 frames after its spawn and a slot
  // respawns in the frame after it terminates; more frames than this bound
  // means paths were lost, which must end the call rather than spin.
  const unsigned long long rounds = (limit + ctx->st.slots - 1) / ctx->st.slots + 1;
  const unsigned long long max_frames = rounds * ((unsigned long long)ctx->params.max_bounces + 2ull) + 8ull;
  unsigned long long frames = 0;
  while (true) {
    for (uint32_t f = 0; f < poll; ++f) run_frame(ctx, limit, first, true);
    frames += poll;
    if ((rc = fetch_counters(ctx)) != MIRT_OK) return rc;
    if (ctx->h_stats[S_FINISHED] - finished0 >= limit) break;
    if (frames > max_frames) return fail(ctx, MIRT_ERR_HIP, "render_spp: the wavefront did not drain");
  }
  ctx->next_sample = first + spp;
  ctx->slots_seeded = false;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_synchronize(mirt_ctx *ctx) try {
  ENTER(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ev_collect(ctx);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_read_output(mirt_ctx *ctx, uint32_t *rgb, size_t n_pixels) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (!rgb || n_pixels != (size_t)ctx->st.width * ctx->st.height)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "read_output: n_pixels must be width*height");
  launch_resolve(launch_cfg(ctx), ctx->st, frame_params(ctx, 0, 0));
  HIP_TRY(ctx, hipMemcpyAsync(rgb, ctx->st.output, n_pixels * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ev_collect(ctx);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_read_accum(mirt_ctx *ctx, uint32_t *sum4, uint32_t *count, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "read_accum: n_sources mismatch");
  if (sum4) {
    // the 64-bit sums the kernels keep, as the reference's 32-bit words (clamped to 2^32 - 1)
    launch_export_sums(launch_cfg(ctx), ctx->st, ctx->d_export4);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(sum4, ctx->d_export4, n_sources * 16, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (count) HIP_TRY(ctx, hipMemcpyAsync(count, ctx->st.count, n_sources * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ev_collect(ctx);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_copy_accum_to_device(mirt_ctx *ctx, void *dev_sum4, void *dev_count, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "copy_accum: n_sources mismatch");
  if (dev_sum4) {
    launch_export_sums(launch_cfg(ctx), ctx->st, static_cast<uint32_t *>(dev_sum4));
    HIP_TRY(ctx, hipGetLastError());
  }
  if (dev_count) HIP_TRY(ctx, hipMemcpyAsync(dev_count, ctx->st.count, n_sources * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_load_accum_from_device(mirt_ctx *ctx, const void *dev_sum4, const void *dev_count, size_t n_sources) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (n_sources != ctx->st.n_sources) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "load_accum: n_sources mismatch");
  if (dev_sum4) {
    launch_import_sums(launch_cfg(ctx), ctx->st, static_cast<const uint32_t *>(dev_sum4));
    HIP_TRY(ctx, hipGetLastError());
  }
  if (dev_count) HIP_TRY(ctx, hipMemcpyAsync(ctx->st.count, dev_count, n_sources * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

static int trace_common(mirt_ctx *ctx, const MirtRay *rays, const float *tmax, size_t n, MirtHitId *out,
                        MirtTraversalCounts *counts, bool any_hit, uint32_t repeats, double *ms_per_launch) {
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  if (n == 0) return MIRT_OK;
  if (!rays || (!out && !ms_per_launch) || (any_hit && !tmax)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace: null pointer");
  if (n > 0xFFFFFFFFull) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace: more than 2^32-1 rays");
  for (size_t i = 0; i < n; ++i)
    if (rays[i].last_inst != 0x7FFFFFFFu && rays[i].last_inst >= ctx->n_instances && rays[i].last_inst != 0xFFFFFFFFu)
      return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace: last_inst out of range");
  if (any_hit)
    for (size_t i = 0; i < n; ++i)
      if (!(tmax[i] >= 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace_any: tmax must be >= 0");
  std::vector<void *> tmp;
  DRay *d_rays = nullptr;
  float *d_tmax = nullptr;
  MirtHitId *d_out = nullptr;
  int rc;
  if ((rc = dev_alloc(ctx, tmp, &d_rays, n)) != MIRT_OK) { free_all(tmp); return rc; }
  if ((rc = dev_alloc(ctx, tmp, &d_out, n)) != MIRT_OK) { free_all(tmp); return rc; }
  if (any_hit && (rc = dev_alloc(ctx, tmp, &d_tmax, n)) != MIRT_OK) { free_all(tmp); return rc; }
  hipError_t e = copy_sync(ctx->stream, d_rays, rays, n * sizeof(MirtRay), hipMemcpyHostToDevice);
  if (e == hipSuccess && any_hit) e = copy_sync(ctx->stream, d_tmax, tmax, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_trace_counts, 0, 6 * sizeof(unsigned long long), ctx->stream);
  if (e != hipSuccess) { free_all(tmp); return fail(ctx, MIRT_ERR_HIP, hipGetErrorString(e)); }
  LaunchCfg cfg;
  cfg.stream = ctx->stream;
  cfg.grid_blocks = ctx->num_cus * 2;
  cfg.num_cus = ctx->num_cus;
  uint32_t *work = reinterpret_cast<uint32_t *>(ctx->d_trace_counts + 5);  // chunk cursor, zeroed before every launch
  if (ms_per_launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    launch_trace(cfg, ctx->sc, d_rays, d_tmax, n, d_out, ctx->d_trace_counts, work, any_hit, false);  // warm-up
    float total_ms = 0.0f;
    for (uint32_t r = 0; r < repeats; ++r) {
      (void)hipMemsetAsync(work, 0, sizeof(uint32_t), ctx->stream);
      (void)hipEventRecord(a, ctx->stream);
      launch_trace(cfg, ctx->sc, d_rays, d_tmax, n, d_out, ctx->d_trace_counts, work, any_hit, false);
      (void)hipEventRecord(b, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);
      float one = 0.0f;
      (void)hipEventElapsedTime(&one, a, b);
      total_ms += one;
    }
    e = hipStreamSynchronize(ctx->stream);
    *ms_per_launch = repeats ? (double)total_ms / repeats : 0.0;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  } else {
    launch_trace(cfg, ctx->sc, d_rays, d_tmax, n, d_out, ctx->d_trace_counts, work, any_hit, counts != nullptr);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess && out) e = copy_sync(ctx->stream, out, d_out, n * sizeof(MirtHitId), hipMemcpyDeviceToHost);
  if (e == hipSuccess && counts) {
    unsigned long long h[5];
    e = copy_sync(ctx->stream, h, ctx->d_trace_counts, sizeof(h), hipMemcpyDeviceToHost);
    counts->rays = h[0]; counts->tlas_nodes = h[1]; counts->instance_visits = h[2];
    counts->blas_nodes = h[3]; counts->triangles = h[4];
  }
  free_all(tmp);
  if (e != hipSuccess) return fail(ctx, MIRT_ERR_HIP, hipGetErrorString(e));
  return MIRT_OK;
}

int mirt_trace_closest(mirt_ctx *ctx, const MirtRay *rays, size_t n, MirtHitId *out, MirtTraversalCounts *counts) try {
  ENTER(ctx);
  return trace_common(ctx, rays, nullptr, n, out, counts, false, 0, nullptr);
} catch (...) {
  return on_exception(ctx);
}

int mirt_trace_closest_static(mirt_ctx *ctx, const MirtRay *rays, size_t n, MirtHitId *out,
                              MirtTraversalCounts *counts) try {
  ENTER(ctx);
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  if (n == 0) return MIRT_OK;
  if (!rays || !out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace: null pointer");
  if (n > kStaticTraceMaxRays)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT,
                "trace_closest_static: at most " + std::to_string(kStaticTraceMaxRays) +
                    " rays per call (the workspace is a fixed array); split the set or use mirt_trace_closest");
  for (size_t i = 0; i < n; ++i)
    if (rays[i].last_inst != 0x7FFFFFFFu && rays[i].last_inst >= ctx->n_instances && rays[i].last_inst != 0xFFFFFFFFu)
      return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "trace: last_inst out of range");
  unsigned char *ws = ctx->static_ws;
  if (!ws) return fail(ctx, MIRT_ERR_HIP, "trace_closest_static: the address of the static workspace is not available on this device");
  unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(ws);
  uint32_t *work = reinterpret_cast<uint32_t *>(d_counts + 5);
  DRay *d_rays = reinterpret_cast<DRay *>(ws + kStaticTraceRayOff);
  MirtHitId *d_out = reinterpret_cast<MirtHitId *>(ws + kStaticTraceOutOff);
  std::lock_guard<std::mutex> one_user(g_scratch_use[ctx->device]);  // static_ws != nullptr implies device < kMaxDevices
  // the array keeps the last call's contents: counters and cursor start at zero,
  // rays and results of this call overwrite the first n records
  HIP_TRY(ctx, hipMemsetAsync(ws, 0, kStaticTraceRayOff, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_rays, rays, n * sizeof(MirtRay), hipMemcpyHostToDevice, ctx->stream));
  LaunchCfg cfg;
  cfg.stream = ctx->stream;
  cfg.grid_blocks = ctx->num_cus * 2;
  cfg.num_cus = ctx->num_cus;
  launch_trace(cfg, ctx->sc, d_rays, nullptr, n, d_out, d_counts, work, false, counts != nullptr);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n * sizeof(MirtHitId), hipMemcpyDeviceToHost, ctx->stream));
  unsigned long long h[5] = {0, 0, 0, 0, 0};
  if (counts) HIP_TRY(ctx, hipMemcpyAsync(h, d_counts, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (counts) {
    counts->rays = h[0]; counts->tlas_nodes = h[1]; counts->instance_visits = h[2];
    counts->blas_nodes = h[3]; counts->triangles = h[4];
  }
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_trace_any(mirt_ctx *ctx, const MirtRay *rays, const float *tmax, size_t n, MirtHitId *out,
                   MirtTraversalCounts *counts) try {
  ENTER(ctx);
  return trace_common(ctx, rays, tmax, n, out, counts, true, 0, nullptr);
} catch (...) {
  return on_exception(ctx);
}

int mirt_bench_trace_closest(mirt_ctx *ctx, const MirtRay *rays, size_t n, uint32_t repeats, double *ms_per_launch) try {
  ENTER(ctx);
  if (!ms_per_launch || repeats == 0) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "bench_trace: bad arguments");
  return trace_common(ctx, rays, nullptr, n, nullptr, nullptr, false, repeats, ms_per_launch);
} catch (...) {
  return on_exception(ctx);
}

int mirt_bench_gather(mirt_ctx *ctx, size_t table_bytes, uint32_t steps, uint32_t repeats, double *lane_steps_per_s) try {
  ENTER(ctx);
  if (!lane_steps_per_s) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "bench_gather: null pointer");
  if (table_bytes == 0) {
    if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "bench_gather: table_bytes 0 means the scene's size, and no scene is uploaded");
    table_bytes = ctx->sc.geo_bytes;
  }
  std::string err;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const int rc = gather_ceiling(ctx->stream, ctx->num_cus, table_bytes, steps, repeats, lane_steps_per_s, err);
  if (rc != MIRT_OK) return fail(ctx, rc, err);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_scene_bytes(mirt_ctx *ctx, uint64_t *node_and_triangle_bytes, uint64_t *leaf_cache_bytes) try {
  ENTER(ctx);
  if (!ctx->have_scene) return fail(ctx, MIRT_ERR_NOT_READY, "no scene uploaded");
  if (node_and_triangle_bytes) *node_and_triangle_bytes = ctx->sc.geo_bytes;
  if (leaf_cache_bytes) *leaf_cache_bytes = ctx->sc.leaf_cache_bytes;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_stats(mirt_ctx *ctx, MirtStats *out) try {
  ENTER(ctx);
  if (!out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "stats: null pointer");
  memset(out, 0, sizeof(*out));
  if (ctx->have_state) {
    // the overflow statistic is a property of the sums as they stand: counted now
    HIP_TRY(ctx, hipMemsetAsync(&ctx->st.stats[S_OVERFLOW], 0, sizeof(unsigned long long), ctx->stream));
    launch_count_saturated(launch_cfg(ctx), ctx->st);
    HIP_TRY(ctx, hipGetLastError());
    int rc = fetch_counters(ctx);
    if (rc != MIRT_OK) return rc;
    const unsigned long long *s = ctx->h_stats;
    out->extend_rays = s[S_EXTEND_RAYS];
    out->connect_rays = s[S_CONNECT_RAYS];
    out->shade_events = s[S_SHADE_EVENTS];
    out->spawned_samples = s[S_SPAWNED];
    out->finished_samples = s[S_FINISHED];
    out->extend_counts.rays = s[S_EXTEND_RAYS];
    out->extend_counts.tlas_nodes = s[S_EXT_TLAS];
    out->extend_counts.instance_visits = s[S_EXT_INST];
    out->extend_counts.blas_nodes = s[S_EXT_BLAS];
    out->extend_counts.triangles = s[S_EXT_TRI];
    out->connect_counts.rays = s[S_CONNECT_RAYS];
    out->connect_counts.tlas_nodes = s[S_CON_TLAS];
    out->connect_counts.instance_visits = s[S_CON_INST];
    out->connect_counts.blas_nodes = s[S_CON_BLAS];
    out->connect_counts.triangles = s[S_CON_TRI];
    out->extend_tri_wave_lds = s[S_EXT_TRI_LDS];
    out->extend_tri_wave_mem = s[S_EXT_TRI_MEM];
    out->connect_tri_wave_lds = s[S_CON_TRI_LDS];
    out->connect_tri_wave_mem = s[S_CON_TRI_MEM];
    out->overflowed_sums = s[S_OVERFLOW];
    for (int k = 0; k < 16; ++k) out->engine_phase[k] = s[S_PHASE0 + k];
  }
  out->frames = ctx->frames_total;
  out->extend_launches = ctx->kernel_launches[K_EXTEND];   out->extend_ms = ctx->kernel_ms[K_EXTEND];
  out->finish_launches = ctx->kernel_launches[K_FINISH];   out->finish_ms = ctx->kernel_ms[K_FINISH];
  out->connect_launches = ctx->kernel_launches[K_CONNECT]; out->connect_ms = ctx->kernel_ms[K_CONNECT];
  out->shade_launches = ctx->kernel_launches[K_SHADE];     out->shade_ms = ctx->kernel_ms[K_SHADE];
  out->sample_launches = ctx->kernel_launches[K_SAMPLE];   out->sample_ms = ctx->kernel_ms[K_SAMPLE];
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_stats_reset(mirt_ctx *ctx) try {
  ENTER(ctx);
  if (ctx->have_state) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    HIP_TRY(ctx, hipMemsetAsync(ctx->st.stats, 0, S_COUNT * sizeof(unsigned long long), ctx->stream));
  }
  for (int k = 0; k < K_KINDS; ++k) { ctx->kernel_ms[k] = 0.0; ctx->kernel_launches[k] = 0; }
  ctx->frames_total = 0;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_set_counting(mirt_ctx *ctx, int enabled) try {
  ENTER(ctx);
  ctx->counting = enabled != 0;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_set_event_timing(mirt_ctx *ctx, int enabled) try {
  ENTER(ctx);
  ctx->event_timing = enabled != 0;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_dump_extension_rays(mirt_ctx *ctx, MirtRay *out, size_t capacity, size_t *n_written) try {
  ENTER(ctx);
  if (!ctx->have_state) return fail(ctx, MIRT_ERR_NOT_READY, "no state");
  if (!out || !n_written) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "dump: null pointer");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the next extend launch pops the extension queue of the current parity
  // after the sample kernel has topped it up; what is there now are the
  // bounce rays pushed by the last shade launch
  const uint32_t par = ctx->frame & 1u;
  uint32_t cnt = 0;
  HIP_TRY(ctx, copy_sync(ctx->stream, &cnt, ctx->st.qcount + Q_EXT0 + par, 4, hipMemcpyDeviceToHost));
  if (cnt > ctx->st.slots) return fail(ctx, MIRT_ERR_HIP, "dump: queue count exceeds slots");
  std::vector<uint32_t> idx(cnt);
  if (cnt) HIP_TRY(ctx, copy_sync(ctx->stream, idx.data(), ctx->st.qdata[Q_EXT0 + par], (size_t)cnt * 4, hipMemcpyDeviceToHost));
  // path-state rays keep position + last instance in their first quarter, direction + last triangle in the second
  std::vector<uint32_t> all((size_t)8 * ctx->st.slots);
  HIP_TRY(ctx, copy_sync(ctx->stream, all.data(), ctx->st.ray, (size_t)ctx->st.slots * sizeof(DRay), hipMemcpyDeviceToHost));
  size_t w = 0;
  for (uint32_t k = 0; k < cnt && w < capacity; ++k) {
    const uint32_t *q = &all[(size_t)8 * idx[k]];
    MirtRay r;
    memcpy(r.pos, q, 12);
    memcpy(r.dir, q + 4, 12);
    r.last_inst = q[3];
    r.last_prim = q[7];
    out[w++] = r;
  }
  *n_written = w;
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_math_selftest(mirt_ctx *ctx, const float *a, const float *b, const float *c, size_t n, float *out6) try {
  ENTER(ctx);
  if (!a || !b || !c || !out6) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "selftest: null pointer");
  if (n == 0) return MIRT_OK;
  std::vector<void *> tmp;
  float *da = nullptr, *db = nullptr, *dc = nullptr, *dout = nullptr;
  int rc;
  if ((rc = dev_alloc(ctx, tmp, &da, n)) != MIRT_OK || (rc = dev_alloc(ctx, tmp, &db, n)) != MIRT_OK ||
      (rc = dev_alloc(ctx, tmp, &dc, n)) != MIRT_OK || (rc = dev_alloc(ctx, tmp, &dout, 6 * n)) != MIRT_OK) {
    free_all(tmp);
    return rc;
  }
  hipError_t e = copy_sync(ctx->stream, da, a, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = copy_sync(ctx->stream, db, b, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = copy_sync(ctx->stream, dc, c, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_math_selftest(ctx->stream, da, db, dc, n, dout);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = copy_sync(ctx->stream, out6, dout, 6 * n * 4, hipMemcpyDeviceToHost);
  free_all(tmp);
  if (e != hipSuccess) return fail(ctx, MIRT_ERR_HIP, hipGetErrorString(e));
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

// ------------------------------------------------------------------ groups

}  // extern "C"

struct mirt_group {
  std::vector<mirt_ctx *> members;
  std::string error;
  uint32_t next_sample = 0;  // the group's sample-number cursor
  // staging for the reduction, on member 0's device: one copy of the accumulators per other member
  std::vector<unsigned long long *> tmp_sum3;
  std::vector<uint32_t *> tmp_count;
  std::vector<unsigned long long *> tmp_moments;  // adaptive states only
  std::vector<hipEvent_t> copied;  // member i's copies into its staging have been issued up to here
  size_t tmp_sources = 0;
};

namespace {

int gfail(mirt_group *g, int status, const std::string &msg) {
  if (g) g->error = msg;
  return status;
}

int on_group_exception(mirt_group *g) noexcept {
  std::string msg;
  const int status = exception_status(msg);
  try {
    if (g) g->error = msg;
  } catch (...) {
  }
  return status;
}

// Run `call` on every member; the first failure ends the loop and is reported
// with the member's own message.
template <class F>
int for_members(mirt_group *g, const char *what, F call) {
  if (!g) return MIRT_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < g->members.size(); ++i) {
    const int rc = call(g->members[i]);
    if (rc != MIRT_OK)
      return gfail(g, rc, std::string(what) + ": member " + std::to_string(i) + ": " + mirt_last_error(g->members[i]));
  }
  return MIRT_OK;
}

void group_free_staging(mirt_group *g) {
  if (g->members.empty()) return;
  if (hipSetDevice(g->members[0]->device) != hipSuccess) return;
  for (unsigned long long *p : g->tmp_sum3) (void)hipFree(p);
  for (uint32_t *p : g->tmp_count) (void)hipFree(p);
  for (unsigned long long *p : g->tmp_moments) (void)hipFree(p);
  g->tmp_sum3.clear();
  g->tmp_count.clear();
  g->tmp_moments.clear();
  g->tmp_sources = 0;
}

}  // namespace

extern "C" {

int mirt_build_blas(mirt_ctx *ctx, const float *positions4, uint32_t n_positions, uint32_t *faces4, uint32_t n_faces,
                    MirtBVHNode *nodes, uint32_t nodes_capacity, uint32_t *n_nodes, float *device_ms) try {
  ENTER(ctx);
  if (!n_nodes || !nodes || nodes_capacity == 0 || (n_faces != 0 && (!faces4 || !positions4)))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "build_blas: null argument");
  if (n_faces > (1u << 30)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "build_blas: too many faces");
  if (device_ms) *device_ms = 0.0f;
  if (n_faces == 0) {
    // BLAS::new keeps the empty root leaf (src/blas.rs:64-71)
    memset(&nodes[0], 0, sizeof(MirtBVHNode));
    nodes[0].is_leaf = 1u;
    *n_nodes = 1;
    return MIRT_OK;
  }
  // the centroid also reads positions[face.w]
  for (size_t k = 0; k < (size_t)n_faces * 4; ++k)
    if (faces4[k] >= n_positions) return fail(ctx, MIRT_ERR_BAD_SCENE, "build_blas: face index out of range");
  std::string err;
  const int rc = build_blas_device(ctx->stream, positions4, n_positions, faces4, n_faces, nodes, nodes_capacity, n_nodes,
                                   device_ms, &ctx->build_ws, &ctx->build_ws_bytes, err);
  if (rc != MIRT_OK) return fail(ctx, rc, err);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_build_tlas(mirt_ctx *ctx, const float *geom_aabbs6, uint32_t n_geometries, const MirtTransform *transforms,
                    uint32_t n_transforms, const MirtInstance *instances, uint32_t n_instances, MirtBVHNode *nodes,
                    uint32_t nodes_capacity, uint32_t *n_nodes, uint32_t *instance_ids, float *device_ms) try {
  ENTER(ctx);
  if (!geom_aabbs6 || !transforms || !instances || !nodes || !n_nodes || !instance_ids || n_instances == 0 || nodes_capacity == 0)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "build_tlas: null argument or no instances");
  if (n_instances > (1u << 30)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "build_tlas: too many instances");
  if (device_ms) *device_ms = 0.0f;
  // M = translate * (Rx * (Ry * Rz)) * scale (src/tlas.rs:80-92) per instance on the host — its sin / cos
  // are the ones the reference's glam calls — as a column-major 4x4
  std::vector<float> mats((size_t)16 * n_instances);
  std::vector<uint32_t> geom(n_instances);
  for (uint32_t i = 0; i < n_instances; ++i)
    if (instances[i].transform >= n_transforms || instances[i].geometry >= n_geometries)
      return fail(ctx, MIRT_ERR_BAD_SCENE, "build_tlas: instance refers to a missing transform or geometry");
  parallel_for(n_instances, [&](uint32_t i) {
    float M[12], Mi[12];
    instance_matrices(transforms[instances[i].transform], M, Mi);
    float *c = &mats[(size_t)16 * i];
    for (int col = 0; col < 4; ++col)
      for (int r = 0; r < 3; ++r) c[4 * col + r] = M[4 * r + col];
    c[3] = c[7] = c[11] = 0.0f;
    c[15] = 1.0f;
    geom[i] = instances[i].geometry;
  });
  std::string err;
  const int rc = build_tlas_device(ctx->stream, geom_aabbs6, n_geometries, mats.data(), geom.data(), n_instances, nodes,
                                   nodes_capacity, n_nodes, instance_ids, device_ms, &ctx->build_ws, &ctx->build_ws_bytes, err);
  if (rc != MIRT_OK) return fail(ctx, rc, err);
  return MIRT_OK;
} catch (...) {
  return on_exception(ctx);
}

int mirt_group_create(const int *device_ordinals, int n, mirt_group **out_group) try {
  if (!device_ordinals || n < 1 || !out_group) return MIRT_ERR_INVALID_ARGUMENT;
  *out_group = nullptr;
  mirt_group *g = new (std::nothrow) mirt_group;
  if (!g) return MIRT_ERR_OUT_OF_MEMORY;
  for (int i = 0; i < n; ++i) {
    mirt_ctx *c = nullptr;
    const int rc = mirt_create(device_ordinals[i], &c);
    if (rc != MIRT_OK) {
      for (mirt_ctx *m : g->members) (void)mirt_destroy(m);
      delete g;
      return rc;
    }
    g->members.push_back(c);
  }
  // direct copies between member 0 and the others where the devices allow it
  // (the reduction works without, through host staging inside the runtime)
  for (int i = 1; i < n; ++i) {
    if (device_ordinals[i] == device_ordinals[0]) continue;
    // both ways: the reduction's copies run on the other member's stream and write into member 0's memory
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device_ordinals[0], device_ordinals[i]) == hipSuccess && can &&
        hipSetDevice(device_ordinals[0]) == hipSuccess)
      (void)hipDeviceEnablePeerAccess(device_ordinals[i], 0);  // "already enabled" is fine
    can = 0;
    if (hipDeviceCanAccessPeer(&can, device_ordinals[i], device_ordinals[0]) == hipSuccess && can &&
        hipSetDevice(device_ordinals[i]) == hipSuccess)
      (void)hipDeviceEnablePeerAccess(device_ordinals[0], 0);
  }
  (void)hipGetLastError();
  *out_group = g;
  return MIRT_OK;
} catch (...) {
  return on_group_exception(nullptr);
}

int mirt_group_destroy(mirt_group *g) try {
  if (!g) return MIRT_ERR_INVALID_ARGUMENT;
  group_free_staging(g);
  for (hipEvent_t ev : g->copied) (void)hipEventDestroy(ev);
  for (mirt_ctx *m : g->members) (void)mirt_destroy(m);
  delete g;
  return MIRT_OK;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_size(const mirt_group *g) { return g ? (int)g->members.size() : 0; }

mirt_ctx *mirt_group_member(mirt_group *g, int i) {
  return (g && i >= 0 && (size_t)i < g->members.size()) ? g->members[(size_t)i] : nullptr;
}

const char *mirt_group_last_error(const mirt_group *g) { return g ? g->error.c_str() : "null group"; }

int mirt_group_scene_upload(mirt_group *g, const MirtSceneDesc *scene) try {
  return for_members(g, "scene_upload", [&](mirt_ctx *c) { return mirt_scene_upload(c, scene); });
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_scene_update_instances(mirt_group *g, const MirtSceneDesc *scene) try {
  return for_members(g, "scene_update_instances", [&](mirt_ctx *c) { return mirt_scene_update_instances(c, scene); });
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_scene_update_transforms(mirt_group *g, const MirtTransform *transforms, uint32_t n_transforms, float *device_ms) try {
  float worst = 0.0f;
  const int rc = for_members(g, "scene_update_transforms", [&](mirt_ctx *c) {
    float ms = 0.0f;
    const int r = mirt_scene_update_transforms(c, transforms, n_transforms, &ms);
    if (ms > worst) worst = ms;
    return r;
  });
  if (device_ms) *device_ms = worst;
  return rc;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_state_create(mirt_group *g, uint32_t width, uint32_t height, uint32_t threads,
                            const MirtSampleSource *sources, uint32_t n_sources, const MirtParams *params) try {
  const int rc = for_members(g, "state_create", [&](mirt_ctx *c) {
    return mirt_state_create(c, width, height, threads, sources, n_sources, params);
  });
  if (rc == MIRT_OK) g->next_sample = 0;
  return rc;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_camera_set(mirt_group *g, const MirtCamera *camera) try {
  const int rc = for_members(g, "camera_set", [&](mirt_ctx *c) { return mirt_camera_set(c, camera); });
  if (rc == MIRT_OK && camera && camera->changed != 0) g->next_sample = 0;  // the members cleared their accumulators
  return rc;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_reset_accum(mirt_group *g) try {
  const int rc = for_members(g, "reset_accum", [&](mirt_ctx *c) { return mirt_reset_accum(c); });
  if (rc == MIRT_OK) g->next_sample = 0;
  return rc;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_render_spp(mirt_group *g, uint32_t spp) try {
  if (!g) return MIRT_ERR_INVALID_ARGUMENT;
  const size_t n = g->members.size();
  std::vector<int> rcs(n, MIRT_OK);
  std::vector<std::thread> workers;
  uint32_t first = g->next_sample;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t mine = spp / (uint32_t)n + (i < spp % n ? 1u : 0u);
    if (mine == 0) continue;
    mirt_ctx *c = g->members[i];
    int *rc = &rcs[i];
    // one host thread per member: mirt_render_spp polls its device until the wavefront has drained
    auto work = [c, first, mine, rc]() {
      *rc = mirt_set_sample_range(c, first);
      if (*rc == MIRT_OK) *rc = mirt_render_spp(c, mine);
    };
    try {
      workers.emplace_back(work);
    } catch (...) {  // no thread to be had: this member renders on the calling thread (nothing is thrown across the C ABI)
      work();
    }
    first += mine;
  }
  for (std::thread &w : workers) w.join();
  for (size_t i = 0; i < n; ++i)
    if (rcs[i] != MIRT_OK)
      return gfail(g, rcs[i], "render_spp: member " + std::to_string(i) + ": " + mirt_last_error(g->members[i]));
  g->next_sample = first;
  return MIRT_OK;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_adapt_update(mirt_group *g, float rel_threshold, uint32_t min_samples, uint32_t *n_active) try {
  if (!g) return MIRT_ERR_INVALID_ARGUMENT;
  // the totals — sums, counts, moments of every member — on member 0, which decides; the others take its flags
  int rc = mirt_group_reduce(g);
  if (rc != MIRT_OK) return rc;
  mirt_ctx *root = g->members[0];
  if (hipSetDevice(root->device) != hipSuccess) return gfail(g, MIRT_ERR_HIP, "adapt_update: hipSetDevice");
  std::vector<uint32_t> flags;
  uint32_t left = 0;
  rc = adapt_update_impl(root, rel_threshold, min_samples, &left, &flags);
  if (rc != MIRT_OK) return gfail(g, rc, std::string("adapt_update: member 0: ") + mirt_last_error(root));
  for (size_t i = 1; i < g->members.size(); ++i) {
    mirt_ctx *c = g->members[i];
    if (hipSetDevice(c->device) != hipSuccess) return gfail(g, MIRT_ERR_HIP, "adapt_update: hipSetDevice");
    uint32_t theirs = 0;
    rc = install_flags(c, flags.data(), nullptr, &theirs);
    if (rc != MIRT_OK) return gfail(g, rc, "adapt_update: member " + std::to_string(i) + ": " + mirt_last_error(c));
  }
  if (n_active) *n_active = left;
  return MIRT_OK;
} catch (...) {
  return on_group_exception(g);
}

int mirt_group_reduce(mirt_group *g) try {
  if (!g) return MIRT_ERR_INVALID_ARGUMENT;
  mirt_ctx *root = g->members[0];
  if (!root->have_state) return gfail(g, MIRT_ERR_NOT_READY, "reduce: no state");
  const size_t ns = root->st.n_sources;
  const size_t n = g->members.size();
  for (size_t i = 1; i < n; ++i)
    if (!g->members[i]->have_state || g->members[i]->st.n_sources != ns)
      return gfail(g, MIRT_ERR_NOT_READY, "reduce: member " + std::to_string(i) + " has no state of the same size");
  const bool with_moments = root->st.moments != nullptr;
  for (size_t i = 1; i < n; ++i)
    if ((g->members[i]->st.moments != nullptr) != with_moments)
      return gfail(g, MIRT_ERR_NOT_READY, "reduce: member " + std::to_string(i) + " differs in adaptive sampling");
  if (n == 1) return MIRT_OK;
#define GROUP_TRY(expr)                                                                  \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return gfail(g, MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
  GROUP_TRY(hipSetDevice(root->device));
  if (g->tmp_sources != ns || g->tmp_sum3.size() != n - 1 || (with_moments && g->tmp_moments.size() != n - 1)) {
    group_free_staging(g);
    GROUP_TRY(hipSetDevice(root->device));
    for (size_t i = 1; i < n; ++i) {
      unsigned long long *a = nullptr;
      uint32_t *b = nullptr;
      GROUP_TRY(hipMalloc(reinterpret_cast<void **>(&a), ns * 8 * kSumStride));
      g->tmp_sum3.push_back(a);
      GROUP_TRY(hipMalloc(reinterpret_cast<void **>(&b), ns * 4));
      g->tmp_count.push_back(b);
      if (with_moments) {
        unsigned long long *m = nullptr;
        GROUP_TRY(hipMalloc(reinterpret_cast<void **>(&m), ns * 16));
        g->tmp_moments.push_back(m);
      }
    }
    g->tmp_sources = ns;
  }
  // (an event belongs to the device that is current when it is created and can only be recorded on a stream
  // of that device: member i's event is made with member i's device current)
  while (g->copied.size() < n - 1) {
    GROUP_TRY(hipSetDevice(g->members[g->copied.size() + 1]->device));
    hipEvent_t ev;
    GROUP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g->copied.push_back(ev);
  }
  // Every member pushes its accumulators to its own staging area on member 0's
  // device, on its own stream (behind whatever it still has in flight), and
  // clears them behind the copy: the n - 1 transfers run side by side over
  // their own links.  Member 0's stream waits for each copy's event and adds.
  for (size_t i = 1; i < n; ++i) {
    mirt_ctx *c = g->members[i];
    GROUP_TRY(hipSetDevice(c->device));
    GROUP_TRY(hipMemcpyPeerAsync(g->tmp_sum3[i - 1], root->device, c->st.sum3, c->device, ns * 8 * kSumStride, c->stream));
    GROUP_TRY(hipMemcpyPeerAsync(g->tmp_count[i - 1], root->device, c->st.count, c->device, ns * 4, c->stream));
    if (with_moments)
      GROUP_TRY(hipMemcpyPeerAsync(g->tmp_moments[i - 1], root->device, c->st.moments, c->device, ns * 16, c->stream));
    GROUP_TRY(hipEventRecord(g->copied[i - 1], c->stream));
    GROUP_TRY(hipMemsetAsync(c->st.sum3, 0, ns * 8 * kSumStride, c->stream));
    GROUP_TRY(hipMemsetAsync(c->st.count, 0, ns * 4, c->stream));
    if (with_moments) GROUP_TRY(hipMemsetAsync(c->st.moments, 0, ns * 16, c->stream));
  }
  GROUP_TRY(hipSetDevice(root->device));
  for (size_t i = 1; i < n; ++i) {
    GROUP_TRY(hipStreamWaitEvent(root->stream, g->copied[i - 1], 0));
    launch_accum_add64(launch_cfg(root), root->st.sum3, g->tmp_sum3[i - 1], ns * kSumStride);  // 64-bit: clamped when handed out
    launch_accum_add(launch_cfg(root), root->st.count, g->tmp_count[i - 1], ns, false);
    if (with_moments) launch_accum_add64(launch_cfg(root), root->st.moments, g->tmp_moments[i - 1], ns * 2);
  }
  GROUP_TRY(hipGetLastError());
  GROUP_TRY(hipStreamSynchronize(root->stream));
  for (size_t i = 1; i < n; ++i) {
    GROUP_TRY(hipSetDevice(g->members[i]->device));
    GROUP_TRY(hipStreamSynchronize(g->members[i]->stream));
  }
#undef GROUP_TRY
  return MIRT_OK;
} catch (...) {
  return on_group_exception(g);
}

}  // extern "C"
