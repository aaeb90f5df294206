// build.hip — the BLAS builder on the device: BLAS::new + BVH::initialize
// (src/blas.rs:18-78, src/bvh.rs:52-178) for gfx950.
//
// The reference builds a BLAS on a host thread with a recursive median split
// whose partition loop is sequential (src/bvh.rs:105-112):
//     while i < j { if centroid(i)[axis] < split { i += 1 } else { swap(i, j); j -= 1 } }
// Triangle ids are positions in the order this loop leaves behind and node ids
// follow the recursion, so a device builder has to reproduce both exactly.
//
// The loop in closed form.  Call an element L when its centroid is below the
// split and R otherwise; let F be the range read from its front and B the range
// read from its back.  The loop tests a run of F that ends with its first R,
// then a run of B that ends with its first L, and so on, until n - 1 elements
// have been tested; the one element left over is never tested.  With
//   FR(m) = front index of the m-th R of F,  BL(m) = back index of the m-th L of B,
// an element at front index p (back index k = n - 1 - p) is
//   tested as part of F  iff  p + (BL(m-1) + 1) < n - 1,  m - 1 = number of R before p,
//   tested as part of B  iff  k + FR(m) + 1    < n - 1,  m - 1 = number of L behind p,
// and ends at position
//   F, L: p          F, R (the m-th): n - 2 - BL(m-1)
//   B, R: p - 1      B, L (the m-th): FR(m)
//   untested: i = number of tested L  (the split point; it always lands right).
// Two prefix counts and two scatter tables give every element its place, so a
// node of any size is partitioned by one thread block with scans.
//
// The tree is grown level by level, each level in one pass per size class of
// node: the few huge nodes by all blocks together, then one 1024-thread block,
// one wave or one thread per node.  Nodes get temporary ids in creation order; the reference's ids — the
// two children pushed side by side, then the whole left subtree, then the right
// one — and the skip links (src/bvh.rs:159-172) follow from subtree sizes in a
// bottom-up and a top-down sweep over the levels.
//
// min / max keep the first operand on ties like the host model's (host.hpp
// fmin2 / fmax2), which decides the sign of a zero bound: reductions carry the
// position with the value so that the first occurrence wins.
#include <float.h>

#include <string>
#include <vector>

#include "kernels.h"

#ifndef MIRT_BUILD_BIG
#define MIRT_BUILD_BIG 512u
#endif

namespace mirt {

namespace {

struct TNode {
  float lb[3], ub[3];
  uint32_t s, e;      // element range
  uint32_t child;     // temporary id of the left child (right = child + 1), 0 = leaf
  uint32_t desc;      // number of proper descendants
  uint32_t idx;       // id in the reference's numbering
  uint32_t base;      // id the first child gets
  uint32_t skip;      // skip link (reference numbering)
  uint32_t pad[3];
};
static_assert(sizeof(TNode) == 64, "TNode is one 64-byte record");

struct HugeNode;

struct BuildBufs {
  const float4 *pos;
  const uint4 *faces;
  uint4 *faces_out;
  float4 *cen, *bmn, *bmx;  // per face id: centroid, box
  uint32_t *perm, *perm2;   // face id at every position
  uint32_t *rl, *lr;        // per position: R count before (bit 31: the element is R); L count behind
  uint32_t *pos_r, *pos_lb; // m-th R from the front -> front index; m-th L from the back -> back index
  TNode *tn;
  uint32_t *n_tmp;          // [0] nodes created so far, [1] / [2] / [3] how many of them got kBigNode / kHugeNode elements or more / more than kTinyNode
  HugeNode *huge;           // the level's huge nodes
  uint32_t *huge_hdr;       // [0] how many, [1] their chunks in all
  uint32_t *chunk_node;     // chunk -> index into huge[]
  uint32_t *chunk_r;        // R's in the chunk, then (after the scan) R's before it in its node
  uint32_t *chunk_lb;       // L's behind the chunk in its node
  uint32_t nf;
  uint32_t leaf_max;        // nodes of up to this many elements stay leaves: 4 for a BLAS, 1 for the TLAS
};

constexpr uint32_t kBlasLeafMax = 4u;   // BLAS::new: initialize(4) (src/blas.rs:75)
constexpr uint32_t kTlasLeafMax = 1u;   // TLAS::new: initialize(1) (src/tlas.rs:130)
constexpr uint32_t kBigNode = MIRT_BUILD_BIG;   // nodes this big get a 1024-thread block
#ifndef MIRT_BUILD_TINY
#define MIRT_BUILD_TINY 16u
#endif
constexpr uint32_t kTinyNode = MIRT_BUILD_TINY;  // nodes up to this size are split by one thread each, with the loop as written
constexpr uint32_t kHugeNode = 16384u; // nodes this big are split by many blocks, kBuildChunk elements each
constexpr uint32_t kBuildChunk = 4096u;     // four elements per thread of a 1024-thread block

// A node that many blocks split together: what they share between kernels.
struct HugeNode {
  uint32_t id, s, n, axis;
  uint32_t kmin, kmax;       // ordered keys of the centroid range on the axis
  uint32_t n_r, n_l;         // R's and L's in the node
  uint32_t i;                // tested L's = the split point
  uint32_t unt;              // face id of the element that is never tested
  uint32_t chunk0, n_chunks; // its chunks in the level's chunk list
  uint32_t pad[4];
  unsigned long long bk[12]; // box keys: left min xyz, left max xyz, right min xyz, right max xyz
};

__device__ __forceinline__ float hmin2(float a, float b) { return b < a ? b : a; }  // first operand on ties
__device__ __forceinline__ float hmax2(float a, float b) { return b > a ? b : a; }

__device__ __forceinline__ uint32_t b_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t b_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// order-preserving key of a float with both zeros equal
__device__ __forceinline__ uint32_t ord_key(float v) {
  const uint32_t k = __float_as_uint(v + 0.0f);
  return (k & 0x80000000u) ? ~k : (k | 0x80000000u);
}

template <int NT>
struct BlockTmp {
  uint32_t w[16];
  uint32_t w4[4][16];
  unsigned long long k[16];
  float f[16];
};

// Exclusive count of flags over four consecutive tiles of NT elements (element
// j * NT + thread of the group): out[j] for this thread's element of tile j,
// `total` for everyone.  One barrier pair per 4 * NT elements.
template <int NT>
__device__ __forceinline__ void block_excl_count4(BlockTmp<NT> &sh, const bool (&flag)[4], uint32_t (&out)[4],
                                                  uint32_t &total) {
  unsigned long long m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = __ballot(flag[j]);
  uint32_t run = 0u;
  if (NT == 64) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      out[j] = run + b_rank(m[j]);
      run += (uint32_t)__popcll(m[j]);
    }
    total = run;
    return;
  }
  const uint32_t wid = threadIdx.x >> 6;
  if (b_lane() == 0u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh.w4[j][wid] = (uint32_t)__popcll(m[j]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t pre = 0u, tot = 0u;
#pragma unroll
    for (uint32_t w = 0; w < NT / 64; ++w) {
      const uint32_t c = sh.w4[j][w];
      pre += w < wid ? c : 0u;
      tot += c;
    }
    out[j] = run + pre + b_rank(m[j]);
    run += tot;
  }
  __syncthreads();
  total = run;
}

template <int NT>
__device__ __forceinline__ uint32_t block_sum(BlockTmp<NT> &sh, uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
  if (NT == 64) return v;
  const uint32_t wid = threadIdx.x >> 6;
  if (b_lane() == 0u) sh.w[wid] = v;
  __syncthreads();
  uint32_t tot = 0u;
#pragma unroll
  for (uint32_t w = 0; w < NT / 64; ++w) tot += sh.w[w];
  __syncthreads();
  return tot;
}

template <int NT, bool MAX>
__device__ __forceinline__ float block_minmax(BlockTmp<NT> &sh, float v) {
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = MAX ? (o > v ? o : v) : (o < v ? o : v);
  }
  if (NT == 64) return v;
  const uint32_t wid = threadIdx.x >> 6;
  if (b_lane() == 0u) sh.f[wid] = v;
  __syncthreads();
  float r = sh.f[0];
#pragma unroll
  for (uint32_t w = 1; w < NT / 64; ++w) {
    const float o = sh.f[w];
    r = MAX ? (o > r ? o : r) : (o < r ? o : r);
  }
  __syncthreads();
  return r;
}

template <int NT>
__device__ __forceinline__ unsigned long long block_min_key(BlockTmp<NT> &sh, unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  if (NT == 64) return v;
  const uint32_t wid = threadIdx.x >> 6;
  if (b_lane() == 0u) sh.k[wid] = v;
  __syncthreads();
  unsigned long long r = sh.k[0];
#pragma unroll
  for (uint32_t w = 1; w < NT / 64; ++w) {
    const unsigned long long o = sh.k[w];
    r = o < r ? o : r;
  }
  __syncthreads();
  return r;
}

// The box of positions [a, b) of the element order (compute_node_bounds of a
// leaf, src/bvh.rs:58-63): per axis the smallest / largest value, and among
// equal values the first in order — what the sequential fold keeps.
template <int NT>
__device__ __forceinline__ void range_bounds(BlockTmp<NT> &sh, const BuildBufs &bb, uint32_t a, uint32_t b, float lb[3],
                                             float ub[3]) {
  unsigned long long kmin[3] = {~0ull, ~0ull, ~0ull}, kmax[3] = {~0ull, ~0ull, ~0ull};
#pragma unroll 4
  for (uint32_t p = a + threadIdx.x; p < b; p += NT) {
    const uint32_t f = bb.perm[p];
    const float4 mn = bb.bmn[f], mx = bb.bmx[f];
    const float vmn[3] = {mn.x, mn.y, mn.z}, vmx[3] = {mx.x, mx.y, mx.z};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const unsigned long long k0 = ((unsigned long long)ord_key(vmn[ax]) << 32) | (p - a);
      const unsigned long long k1 = ((unsigned long long)(~ord_key(vmx[ax])) << 32) | (p - a);
      kmin[ax] = k0 < kmin[ax] ? k0 : kmin[ax];
      kmax[ax] = k1 < kmax[ax] ? k1 : kmax[ax];
    }
  }
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const uint32_t pmin = (uint32_t)block_min_key<NT>(sh, kmin[ax]);
    const uint32_t pmax = (uint32_t)block_min_key<NT>(sh, kmax[ax]);
    const float4 mn = bb.bmn[bb.perm[a + pmin]], mx = bb.bmx[bb.perm[a + pmax]];
    lb[ax] = ax == 0 ? mn.x : (ax == 1 ? mn.y : mn.z);
    ub[ax] = ax == 0 ? mx.x : (ax == 1 ? mx.y : mx.z);
  }
}

// elem_bounds / elem_centroid of every face (src/blas.rs:19-37) and the
// identity order.  The centroid sums FOUR positions — the face's three and
// positions[face.w = 0] — and divides by three.
__global__ __launch_bounds__(256) void k_build_prep(BuildBufs bb) {
  const uint32_t f = blockIdx.x * 256u + threadIdx.x;
  if (f >= bb.nf) return;
  const uint4 face = bb.faces[f];
  const float4 p0 = bb.pos[face.x], p1 = bb.pos[face.y], p2 = bb.pos[face.z], p3 = bb.pos[face.w];
  bb.bmn[f] = make_float4(hmin2(hmin2(p0.x, p1.x), p2.x), hmin2(hmin2(p0.y, p1.y), p2.y), hmin2(hmin2(p0.z, p1.z), p2.z), 0.0f);
  bb.bmx[f] = make_float4(hmax2(hmax2(p0.x, p1.x), p2.x), hmax2(hmax2(p0.y, p1.y), p2.y), hmax2(hmax2(p0.z, p1.z), p2.z), 0.0f);
  bb.cen[f] = make_float4((((p0.x + p1.x) + p2.x) + p3.x) / 3.0f, (((p0.y + p1.y) + p2.y) + p3.y) / 3.0f,
                          (((p0.z + p1.z) + p2.z) + p3.z) / 3.0f, 0.0f);
  bb.perm[f] = f;
}

// TLAS::new's elements (src/tlas.rs:61-99): the world box of every instance from the eight
// corners of its geometry's box through M = translate * (Rx * (Ry * Rz)) * scale — M comes
// from the host, column-major, built there with the host's own sin / cos so that it has the
// host builder's bits — each corner as x*col0 + y*col1 + z*col2 + w*col3 summed left to
// right, united in the order x, y, z (z fastest); elem_centroid = (lb + ub) / 2 (src/tlas.rs:32-34).
__global__ __launch_bounds__(256) void k_tlas_prep(BuildBufs bb, const float *geom_aabbs6, const float *mats16,
                                                   const uint32_t *inst_geometry) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= bb.nf) return;
  const float *a = geom_aabbs6 + 6 * (size_t)inst_geometry[e];
  const float *m = mats16 + 16 * (size_t)e;
  float lb[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
  bool first = true;
  for (int x = 0; x <= 1; ++x)
    for (int y = 0; y <= 1; ++y)
      for (int z = 0; z <= 1; ++z) {
        const float v[4] = {x == 0 ? a[0] : a[3], y == 0 ? a[1] : a[4], z == 0 ? a[2] : a[5], 1.0f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float acc = m[r] * v[0];
          acc = acc + m[4 + r] * v[1];
          acc = acc + m[8 + r] * v[2];
          acc = acc + m[12 + r] * v[3];
          lb[r] = first ? acc : hmin2(lb[r], acc);
          ub[r] = first ? acc : hmax2(ub[r], acc);
        }
        first = false;
      }
  bb.bmn[e] = make_float4(lb[0], lb[1], lb[2], 0.0f);
  bb.bmx[e] = make_float4(ub[0], ub[1], ub[2], 0.0f);
  bb.cen[e] = make_float4((lb[0] + ub[0]) / 2.0f, (lb[1] + ub[1]) / 2.0f, (lb[2] + ub[2]) / 2.0f, 0.0f);
  bb.perm[e] = e;
}

// the root: a leaf over every face (src/blas.rs:64-71) with its box; the box keys of
// kBuildChunk faces per block meet in root_keys (all ones before the launch)
__global__ __launch_bounds__(1024) void k_build_root_keys(BuildBufs bb, unsigned long long *root_keys) {
  __shared__ BlockTmp<1024> sh;
  unsigned long long k[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) k[q] = ~0ull;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = blockIdx.x * kBuildChunk + (uint32_t)j * 1024u + threadIdx.x;
    if (p >= bb.nf) continue;
    const float4 mn = bb.bmn[p], mx = bb.bmx[p];  // the order is still the identity
    const float vmn[3] = {mn.x, mn.y, mn.z}, vmx[3] = {mx.x, mx.y, mx.z};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const unsigned long long k0 = ((unsigned long long)ord_key(vmn[ax]) << 32) | p;
      const unsigned long long k1 = ((unsigned long long)(~ord_key(vmx[ax])) << 32) | p;
      k[ax] = k0 < k[ax] ? k0 : k[ax];
      k[3 + ax] = k1 < k[3 + ax] ? k1 : k[3 + ax];
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const unsigned long long m = block_min_key<1024>(sh, k[q]);
    if (threadIdx.x == 0u && m != ~0ull) atomicMin(&root_keys[q], m);
  }
}
__global__ void k_build_root(BuildBufs bb, const unsigned long long *root_keys) {
  if (blockIdx.x != 0u || threadIdx.x != 0u) return;
  TNode n{};
  for (int ax = 0; ax < 3; ++ax) {
    const float4 mn = bb.bmn[(uint32_t)root_keys[ax]], mx = bb.bmx[(uint32_t)root_keys[3 + ax]];
    n.lb[ax] = ax == 0 ? mn.x : (ax == 1 ? mn.y : mn.z);
    n.ub[ax] = ax == 0 ? mx.x : (ax == 1 ? mx.y : mx.z);
  }
  n.s = 0u;
  n.e = bb.nf;
  bb.tn[0] = n;
  bb.n_tmp[0] = 1u;
  bb.n_tmp[1] = bb.nf >= kBigNode ? 1u : 0u;
  bb.n_tmp[2] = bb.nf >= kHugeNode ? 1u : 0u;
  bb.n_tmp[3] = bb.nf > kTinyNode ? 1u : 0u;
}

// the longest axis of a node's box (src/bvh.rs:81-88)
__device__ __forceinline__ int split_axis(const TNode &node) {
  const float ex = node.ub[0] - node.lb[0], ey = node.ub[1] - node.lb[1], ez = node.ub[2] - node.lb[2];
  int axis = 0;
  if (ey > ex) axis = 1;
  if (ez > (axis == 0 ? ex : ey)) axis = 2;
  return axis;
}

// Where the partition loop leaves the element at front index p of a node with
// n elements, n_r R's and n_l L's (closed form at the top of this file).
// Returns false for the one element that is never tested; `tested_l` says
// whether a placed element counts towards the split point.
__device__ __forceinline__ bool place_element(const BuildBufs &bb, uint32_t s, uint32_t n, uint32_t n_r, uint32_t n_l,
                                              uint32_t p, uint32_t &dest, bool &tested_l) {
  const unsigned long long kInf = 1ull << 40;
  const uint32_t w = bb.rl[s + p];
  const bool is_r = (w & 0x80000000u) != 0u;
  const uint32_t m1 = w & 0x7FFFFFFFu;  // R's before p
  const unsigned long long prev_b = m1 == 0u ? 0ull : (m1 - 1u < n_l ? (unsigned long long)bb.pos_lb[s + m1 - 1u] + 1ull : kInf);
  const uint32_t k = n - 1u - p, ml = bb.lr[s + p];  // L's behind p
  const unsigned long long fr = ml < n_r ? (unsigned long long)bb.pos_r[s + ml] : kInf;
  const bool t_f = (unsigned long long)p + prev_b < (unsigned long long)(n - 1u);
  const bool t_b = (unsigned long long)k + fr + 1ull < (unsigned long long)(n - 1u);
  tested_l = (t_f || t_b) && !is_r;
  if (t_f) dest = is_r ? n - 1u - (uint32_t)prev_b : p;
  else if (t_b) dest = is_r ? p - 1u : (uint32_t)fr;
  return t_f || t_b;
}

// The two children of node `id` split at s + i (src/bvh.rs:121-137), as leaves with their boxes.
__device__ __forceinline__ void make_children(const BuildBufs &bb, uint32_t id, uint32_t s, uint32_t n, uint32_t i,
                                              const float llb[3], const float lub[3], const float rlb[3],
                                              const float rub[3]) {
  const uint32_t c = atomicAdd(&bb.n_tmp[0], 2u);
  const uint32_t big = (i >= kBigNode ? 1u : 0u) + (n - i >= kBigNode ? 1u : 0u);
  if (big != 0u) atomicAdd(&bb.n_tmp[1], big);
  const uint32_t huge = (i >= kHugeNode ? 1u : 0u) + (n - i >= kHugeNode ? 1u : 0u);
  if (huge != 0u) atomicAdd(&bb.n_tmp[2], huge);
  const uint32_t wave = (i > kTinyNode ? 1u : 0u) + (n - i > kTinyNode ? 1u : 0u);
  if (wave != 0u) atomicAdd(&bb.n_tmp[3], wave);
  TNode l{}, r{};
  for (int ax = 0; ax < 3; ++ax) {
    l.lb[ax] = llb[ax];
    l.ub[ax] = lub[ax];
    r.lb[ax] = rlb[ax];
    r.ub[ax] = rub[ax];
  }
  l.s = s;
  l.e = s + i;
  r.s = s + i;
  r.e = s + n;
  bb.tn[c] = l;
  bb.tn[c + 1u] = r;
  bb.tn[id].child = c;
}

// subdivide (src/bvh.rs:68-157) for the nodes [first, first + gridDim.x) of one
// level, one block per node; the block shape takes the nodes of its size class.
template <int NT>
__global__ __launch_bounds__(NT) void k_build_split(BuildBufs bb, uint32_t first) {
  __shared__ BlockTmp<NT> sh;
  __shared__ uint32_t sh_unt;
  const uint32_t id = first + blockIdx.x;
  const TNode node = bb.tn[id];
  const uint32_t s = node.s, n = node.e - node.s;
  if (n <= bb.leaf_max) return;
  if ((NT == 1024) != (n >= kBigNode) || n >= kHugeNode || n <= kTinyNode) return;
  const uint32_t tid = threadIdx.x;

  const int axis = split_axis(node);
  auto centroid = [&](uint32_t p) {
    const float4 c = bb.cen[bb.perm[s + p]];
    return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
  };

  // split = midpoint of the centroids' range on that axis (src/bvh.rs:94-103)
  float cmin = FLT_MAX, cmax = -FLT_MAX;
#pragma unroll 4
  for (uint32_t p = tid; p < n; p += NT) {
    const float c = centroid(p);
    cmin = c < cmin ? c : cmin;
    cmax = c > cmax ? c : cmax;
  }
  cmin = block_minmax<NT, false>(sh, cmin);
  cmax = block_minmax<NT, true>(sh, cmax);
  const float split = (cmin + cmax) * 0.5f;

  // from the front: R count before every element, table of the R's
  uint32_t n_r = 0u;
  for (uint32_t base = 0; base < n; base += 4u * NT) {
    bool is_r[4];
    uint32_t ex[4], tot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t p = base + (uint32_t)j * NT + tid;
      is_r[j] = p < n && !(centroid(p) < split);
    }
    block_excl_count4<NT>(sh, is_r, ex, tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t p = base + (uint32_t)j * NT + tid;
      if (p < n) bb.rl[s + p] = (n_r + ex[j]) | (is_r[j] ? 0x80000000u : 0u);
      if (is_r[j]) bb.pos_r[s + n_r + ex[j]] = p;
    }
    n_r += tot;
  }
  __syncthreads();
  // from the back: L count behind every element, table of the L's
  uint32_t n_l = 0u;
  for (uint32_t base = 0; base < n; base += 4u * NT) {
    bool is_l[4];
    uint32_t ex[4], tot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t k = base + (uint32_t)j * NT + tid;
      is_l[j] = k < n && (bb.rl[s + n - 1u - k] & 0x80000000u) == 0u;
    }
    block_excl_count4<NT>(sh, is_l, ex, tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t k = base + (uint32_t)j * NT + tid;
      if (k < n) bb.lr[s + n - 1u - k] = n_l + ex[j];
      if (is_l[j]) bb.pos_lb[s + n_l + ex[j]] = k;
    }
    n_l += tot;
  }
  __syncthreads();

  // every tested element to its place; count the tested L's
  uint32_t my_l = 0u;
  if (tid == 0u) sh_unt = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t *__restrict__ perm_in = bb.perm;
  uint32_t *__restrict__ perm_out = bb.perm2;
#pragma unroll 4
  for (uint32_t p = tid; p < n; p += NT) {
    uint32_t d = 0u;
    bool tl = false;
    const uint32_t f = perm_in[s + p];
    if (place_element(bb, s, n, n_r, n_l, p, d, tl)) {
      perm_out[s + d] = f;
      my_l += tl ? 1u : 0u;
    } else {
      sh_unt = f;  // exactly one element is never tested
    }
  }
  const uint32_t i = block_sum<NT>(sh, my_l);  // the split point, relative to s
  __syncthreads();
  if (tid == 0u) bb.perm2[s + i] = sh_unt;
  __syncthreads();
  // the swaps happen whether or not the node ends up split
  {
    const uint32_t *__restrict__ src = bb.perm2;
    uint32_t *__restrict__ dst = bb.perm;
#pragma unroll 4
    for (uint32_t p = tid; p < n; p += NT) dst[s + p] = src[s + p];
  }
  __syncthreads();
  if (i == 0u) return;  // one-sided: the node stays a leaf (src/bvh.rs:114-119; i == n cannot happen)

  float llb[3], lub[3], rlb[3], rub[3];
  range_bounds<NT>(sh, bb, s, s + i, llb, lub);
  range_bounds<NT>(sh, bb, s + i, s + n, rlb, rub);
  if (tid == 0u) make_children(bb, id, s, n, i, llb, lub, rlb, rub);
}

// subdivide for the nodes of up to kTinyNode elements, one thread per node: the
// loops of src/bvh.rs:94-112 and :58-63 as they are written.  Deep levels hold
// hundreds of thousands of such nodes; a wave apiece would mostly idle.
__global__ __launch_bounds__(256) void k_build_split_tiny(BuildBufs bb, uint32_t first, uint32_t count) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  const uint32_t id = first + t;
  const TNode node = bb.tn[id];
  const uint32_t s = node.s, e = node.e, n = e - s;
  if (n <= bb.leaf_max || n > kTinyNode) return;
  const int axis = split_axis(node);
  auto centroid = [&](uint32_t p) {
    const float4 c = bb.cen[bb.perm[p]];
    return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
  };
  float cmin = centroid(s), cmax = cmin;
  for (uint32_t p = s + 1u; p < e; ++p) {
    const float c = centroid(p);
    cmin = hmin2(cmin, c);
    cmax = hmax2(cmax, c);
  }
  const float split = (cmin + cmax) * 0.5f;
  uint32_t i = s, j = e - 1u;
  while (i < j) {
    if (centroid(i) < split) {
      i += 1u;
    } else {
      const uint32_t a = bb.perm[i], b = bb.perm[j];
      bb.perm[i] = b;
      bb.perm[j] = a;
      j -= 1u;
    }
  }
  if (i == s) return;  // one-sided (i == e cannot happen)
  float box[12];
  for (int side = 0; side < 2; ++side) {
    const uint32_t a = side == 0 ? s : i, b = side == 0 ? i : e;
    float4 mn = bb.bmn[bb.perm[a]], mx = bb.bmx[bb.perm[a]];
    for (uint32_t p = a + 1u; p < b; ++p) {
      const float4 m2 = bb.bmn[bb.perm[p]], x2 = bb.bmx[bb.perm[p]];
      mn = make_float4(hmin2(mn.x, m2.x), hmin2(mn.y, m2.y), hmin2(mn.z, m2.z), 0.0f);
      mx = make_float4(hmax2(mx.x, x2.x), hmax2(mx.y, x2.y), hmax2(mx.z, x2.z), 0.0f);
    }
    box[side * 6 + 0] = mn.x; box[side * 6 + 1] = mn.y; box[side * 6 + 2] = mn.z;
    box[side * 6 + 3] = mx.x; box[side * 6 + 4] = mx.y; box[side * 6 + 5] = mx.z;
  }
  make_children(bb, id, s, n, i - s, &box[0], &box[3], &box[6], &box[9]);
}

// ---------------------------------------------------------------- huge nodes
// A node of kHugeNode elements or more would keep one block — one CU — busy for
// milliseconds, so the few such nodes of a level are split by all blocks
// together, kBuildChunk elements per block, in a sequence of short kernels that
// meet in the node's HugeNode record.  The arithmetic is that of k_build_split.

__device__ __forceinline__ float key_to_float(uint32_t k) {  // inverse of ord_key (a zero comes back as +0)
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// The level's huge nodes and their chunks (one block).
__global__ __launch_bounds__(1024) void k_huge_list(BuildBufs bb, uint32_t first, uint32_t count) {
  __shared__ uint32_t sh_n;
  if (threadIdx.x == 0u) sh_n = 0u;
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < count; t += 1024u) {
    const TNode node = bb.tn[first + t];
    const uint32_t n = node.e - node.s;
    if (n >= kHugeNode) {
      const uint32_t b = atomicAdd(&sh_n, 1u);
      HugeNode h{};
      h.id = first + t;
      h.s = node.s;
      h.n = n;
      h.axis = (uint32_t)split_axis(node);
      h.kmin = 0xFFFFFFFFu;
      h.kmax = 0u;
      h.n_chunks = (n + kBuildChunk - 1u) / kBuildChunk;
      for (int q = 0; q < 12; ++q) h.bk[q] = ~0ull;
      bb.huge[b] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0u) {
    uint32_t c = 0u;
    for (uint32_t b = 0; b < sh_n; ++b) {
      bb.huge[b].chunk0 = c;
      c += bb.huge[b].n_chunks;
    }
    bb.huge_hdr[0] = sh_n;
    bb.huge_hdr[1] = c;
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < sh_n; b += 1024u) {
    const uint32_t c0 = bb.huge[b].chunk0, nc = bb.huge[b].n_chunks;
    for (uint32_t c = 0; c < nc; ++c) bb.chunk_node[c0 + c] = b;
  }
}

struct ChunkRef {
  uint32_t b, s, n, lo, hi, axis;  // huge node, its range, this chunk's element range [lo, hi) within it
};
__device__ __forceinline__ bool chunk_ref(const BuildBufs &bb, ChunkRef &c) {
  if (blockIdx.x >= bb.huge_hdr[1]) return false;
  c.b = bb.chunk_node[blockIdx.x];
  const HugeNode &h = bb.huge[c.b];
  c.s = h.s;
  c.n = h.n;
  c.axis = h.axis;
  c.lo = (blockIdx.x - h.chunk0) * kBuildChunk;
  c.hi = c.lo + kBuildChunk < h.n ? c.lo + kBuildChunk : h.n;
  return true;
}
__device__ __forceinline__ float chunk_centroid(const BuildBufs &bb, const ChunkRef &c, uint32_t p) {
  const float4 v = bb.cen[bb.perm[c.s + p]];
  return c.axis == 0u ? v.x : (c.axis == 1u ? v.y : v.z);
}

// centroid range on the axis (src/bvh.rs:94-101)
__global__ __launch_bounds__(1024) void k_huge_range(BuildBufs bb) {
  __shared__ BlockTmp<1024> sh;
  ChunkRef c;
  if (!chunk_ref(bb, c)) return;
  float cmin = FLT_MAX, cmax = -FLT_MAX;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    if (p < c.hi) {
      const float v = chunk_centroid(bb, c, p);
      cmin = v < cmin ? v : cmin;
      cmax = v > cmax ? v : cmax;
    }
  }
  cmin = block_minmax<1024, false>(sh, cmin);
  cmax = block_minmax<1024, true>(sh, cmax);
  if (threadIdx.x == 0u) {
    atomicMin(&bb.huge[c.b].kmin, ord_key(cmin));
    atomicMax(&bb.huge[c.b].kmax, ord_key(cmax));
  }
}

__device__ __forceinline__ float huge_split(const HugeNode &h) {  // src/bvh.rs:103
  return (key_to_float(h.kmin) + key_to_float(h.kmax)) * 0.5f;
}

// R's per chunk
__global__ __launch_bounds__(1024) void k_huge_count(BuildBufs bb) {
  __shared__ BlockTmp<1024> sh;
  ChunkRef c;
  if (!chunk_ref(bb, c)) return;
  const float split = huge_split(bb.huge[c.b]);
  uint32_t mine = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    if (p < c.hi && !(chunk_centroid(bb, c, p) < split)) mine++;
  }
  const uint32_t tot = block_sum<1024>(sh, mine);
  if (threadIdx.x == 0u) bb.chunk_r[blockIdx.x] = tot;
}

// per node: R's before and L's behind every chunk, totals (one thread per huge node)
__global__ __launch_bounds__(256) void k_huge_scan(BuildBufs bb) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= bb.huge_hdr[0]) return;
  HugeNode &h = bb.huge[b];
  uint32_t run = 0u;
  for (uint32_t c = 0; c < h.n_chunks; ++c) {
    const uint32_t r = bb.chunk_r[h.chunk0 + c];
    const uint32_t size = (c + 1u) * kBuildChunk <= h.n ? kBuildChunk : h.n - c * kBuildChunk;
    bb.chunk_r[h.chunk0 + c] = run;
    bb.chunk_lb[h.chunk0 + c] = size - r;  // L's in the chunk, for the pass below
    run += r;
  }
  h.n_r = run;
  uint32_t behind = 0u;
  for (uint32_t c = h.n_chunks; c-- > 0u;) {
    const uint32_t l = bb.chunk_lb[h.chunk0 + c];
    bb.chunk_lb[h.chunk0 + c] = behind;
    behind += l;
  }
  h.n_l = behind;
  h.i = 0u;
  h.unt = 0xFFFFFFFFu;
}

// the two prefix counts and scatter tables of the chunk's elements
__global__ __launch_bounds__(1024) void k_huge_tables(BuildBufs bb) {
  __shared__ BlockTmp<1024> sh;
  ChunkRef c;
  if (!chunk_ref(bb, c)) return;
  const float split = huge_split(bb.huge[c.b]);
  const uint32_t r_before = bb.chunk_r[blockIdx.x], l_behind = bb.chunk_lb[blockIdx.x];
  bool is_r[4];
  uint32_t ex[4], tot;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    is_r[j] = p < c.hi && !(chunk_centroid(bb, c, p) < split);
  }
  block_excl_count4<1024>(sh, is_r, ex, tot);
  const uint32_t l_chunk = (c.hi - c.lo) - tot;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    if (p >= c.hi) continue;
    bb.rl[c.s + p] = (r_before + ex[j]) | (is_r[j] ? 0x80000000u : 0u);
    // L's of the chunk behind p = L's of the chunk - L's up to and including p
    const uint32_t l_upto = (p - c.lo + 1u) - (ex[j] + (is_r[j] ? 1u : 0u));
    const uint32_t lr = l_behind + l_chunk - l_upto;
    bb.lr[c.s + p] = lr;
    if (is_r[j]) bb.pos_r[c.s + r_before + ex[j]] = p;
    else bb.pos_lb[c.s + lr] = c.n - 1u - p;
  }
}

// every tested element to its place; tested L's counted into the node's split point
__global__ __launch_bounds__(1024) void k_huge_place(BuildBufs bb) {
  __shared__ BlockTmp<1024> sh;
  ChunkRef c;
  if (!chunk_ref(bb, c)) return;
  const uint32_t n_r = bb.huge[c.b].n_r, n_l = bb.huge[c.b].n_l;
  uint32_t my_l = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    if (p >= c.hi) continue;
    uint32_t d = 0u;
    bool tl = false;
    const uint32_t f = bb.perm[c.s + p];
    if (place_element(bb, c.s, c.n, n_r, n_l, p, d, tl)) {
      bb.perm2[c.s + d] = f;
      my_l += tl ? 1u : 0u;
    } else {
      bb.huge[c.b].unt = f;
    }
  }
  const uint32_t tot = block_sum<1024>(sh, my_l);
  if (threadIdx.x == 0u && tot != 0u) atomicAdd(&bb.huge[c.b].i, tot);
}

// the new order back into place (the untested element at the split point) and the
// box keys of the two sides
__global__ __launch_bounds__(1024) void k_huge_copy(BuildBufs bb) {
  __shared__ BlockTmp<1024> sh;
  ChunkRef c;
  if (!chunk_ref(bb, c)) return;
  const uint32_t i = bb.huge[c.b].i, unt = bb.huge[c.b].unt;
  unsigned long long k[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) k[q] = ~0ull;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = c.lo + (uint32_t)j * 1024u + threadIdx.x;
    if (p >= c.hi) continue;
    const uint32_t f = p == i ? unt : bb.perm2[c.s + p];
    bb.perm[c.s + p] = f;
    if (i == 0u) continue;  // one-sided: no children, no boxes
    const float4 mn = bb.bmn[f], mx = bb.bmx[f];
    const float vmn[3] = {mn.x, mn.y, mn.z}, vmx[3] = {mx.x, mx.y, mx.z};
    const bool right = p >= i;
    const uint32_t rel = right ? p - i : p;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const unsigned long long k0 = ((unsigned long long)ord_key(vmn[ax]) << 32) | rel;
      const unsigned long long k1 = ((unsigned long long)(~ord_key(vmx[ax])) << 32) | rel;
      if (!right) {
        k[ax] = k0 < k[ax] ? k0 : k[ax];
        k[3 + ax] = k1 < k[3 + ax] ? k1 : k[3 + ax];
      } else {
        k[6 + ax] = k0 < k[6 + ax] ? k0 : k[6 + ax];
        k[9 + ax] = k1 < k[9 + ax] ? k1 : k[9 + ax];
      }
    }
  }
  if (i == 0u) return;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const unsigned long long m = block_min_key<1024>(sh, k[q]);
    if (threadIdx.x == 0u && m != ~0ull) atomicMin(&bb.huge[c.b].bk[q], m);
  }
}

// children of the huge nodes that were split (one thread per node)
__global__ __launch_bounds__(256) void k_huge_finish(BuildBufs bb) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= bb.huge_hdr[0]) return;
  const HugeNode h = bb.huge[b];
  if (h.i == 0u) return;  // stays a leaf (src/bvh.rs:114-119)
  float box[12];
  for (int q = 0; q < 12; ++q) {
    const bool right = q >= 6, is_max = (q % 6) >= 3;
    const int ax = q % 3;
    const uint32_t pos = (uint32_t)h.bk[q] + (right ? h.i : 0u);
    const uint32_t f = bb.perm[h.s + pos];
    const float4 v = is_max ? bb.bmx[f] : bb.bmn[f];
    box[q] = ax == 0 ? v.x : (ax == 1 ? v.y : v.z);
  }
  make_children(bb, h.id, h.s, h.n, h.i, &box[0], &box[3], &box[6], &box[9]);
}

// Bottom-up over one level: subtree sizes, and the box of an inner node as the
// union of its children's final boxes (src/bvh.rs:54-57,156).
__global__ __launch_bounds__(256) void k_build_up(BuildBufs bb, uint32_t first, uint32_t count) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  TNode n = bb.tn[first + t];
  if (n.child == 0u) {
    n.desc = 0u;
  } else {
    const TNode l = bb.tn[n.child], r = bb.tn[n.child + 1u];
    for (int ax = 0; ax < 3; ++ax) {
      n.lb[ax] = hmin2(l.lb[ax], r.lb[ax]);
      n.ub[ax] = hmax2(l.ub[ax], r.ub[ax]);
    }
    n.desc = 2u + l.desc + r.desc;
  }
  bb.tn[first + t] = n;
}

// Top-down over one level: the reference's node ids (children side by side,
// then the left subtree, then the right one: src/bvh.rs:136-142) and skip links
// (left -> right, right -> the parent's: src/bvh.rs:159-172); writes the node
// in BVHNodeGPU form (src/bvh.rs:200-224).
__global__ __launch_bounds__(256) void k_build_down(BuildBufs bb, uint32_t first, uint32_t count, MirtBVHNode *out) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  const uint32_t id = first + t;
  TNode n = bb.tn[id];
  if (id == 0u) {
    n.idx = 0u;
    n.base = 1u;
    n.skip = 0u;
  }
  uint32_t left = 0u;
  if (n.child != 0u) {
    TNode *l = &bb.tn[n.child], *r = &bb.tn[n.child + 1u];
    left = n.base;
    l->idx = n.base;
    r->idx = n.base + 1u;
    l->base = n.base + 2u;
    r->base = n.base + 2u + l->desc;
    l->skip = n.base + 1u;
    r->skip = n.skip;
  }
  MirtBVHNode g;
  g.lb[0] = n.lb[0]; g.lb[1] = n.lb[1]; g.lb[2] = n.lb[2]; g.lb[3] = 0.0f;
  g.ub[0] = n.ub[0]; g.ub[1] = n.ub[1]; g.ub[2] = n.ub[2]; g.ub[3] = 0.0f;
  g.left = left;
  g.right = n.skip;
  g.is_leaf = n.child == 0u ? 1u : 0u;
  g.start = n.s;
  g.end = n.e;
  g.pad[0] = g.pad[1] = g.pad[2] = 0u;
  out[n.idx] = g;
}

// mesh.faces in the order the build left them in (elem_swap, src/blas.rs:39-41)
__global__ __launch_bounds__(256) void k_build_faces(BuildBufs bb) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p < bb.nf) bb.faces_out[p] = bb.faces[bb.perm[p]];
}

// The TLAS as the traversal kernels read it, written where it lives (layout.h: 32-byte nodes,
// every leaf tagged "size out of line" so that the walk hands it to its service point, the size at
// tlas_big[first slot], the instance of every slot in tlas_to_inst) — from the builder's output on
// the device, without a trip through the host.
__global__ __launch_bounds__(256) void k_pack_tlas(const MirtBVHNode *src, uint32_t n_nodes, const uint32_t *perm,
                                                   uint32_t n_slots, DNode *dst, uint32_t *tlas_big, uint32_t *tlas_to_inst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_slots) tlas_to_inst[i] = perm[i];
  if (i >= n_nodes) return;
  const MirtBVHNode s = src[i];
  DNode d;
  d.lbx = s.lb[0]; d.lby = s.lb[1]; d.lbz = s.lb[2];
  d.ubx = s.ub[0]; d.uby = s.ub[1]; d.ubz = s.ub[2];
  if (s.is_leaf == 1u) {
    d.a = s.start;
    d.b = (s.right & kSkipMask) | (kBigLeaf << 24);
    tlas_big[s.start] = s.end - s.start;
  } else {
    d.a = s.left;
    d.b = s.right & kSkipMask;
  }
  dst[i] = d;
}

// Carves 256-byte aligned arrays out of one workspace allocation.
struct Carver {
  unsigned char *base;
  size_t used = 0;
  template <typename T>
  void take(T **out, size_t count) {
    if (base) *out = reinterpret_cast<T *>(base + used);
    used += ((count ? count : 1) * sizeof(T) + 255u) & ~(size_t)255u;
  }
};

}  // namespace

#define BUILD_TRY(expr)                                              \
  do {                                                               \
    const hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) {                                          \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);       \
      return _e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_HIP; \
    }                                                                \
  } while (0)

// The tree over the prepared elements (boxes, centroids, identity order in bb): root, level
// loop, then the reference's numbering and skip links into d_out.  `n_elems` = bb.nf.
static int grow_tree(hipStream_t stream, BuildBufs &bb, uint32_t n_faces, size_t cap, size_t max_huge, size_t max_chunks,
                     unsigned long long *root_keys, MirtBVHNode *d_out, uint32_t *total_out, std::string &err) {
  BUILD_TRY(hipMemsetAsync(root_keys, 0xFF, 6 * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_build_root_keys, dim3((n_faces + kBuildChunk - 1u) / kBuildChunk), dim3(1024), 0, stream, bb, root_keys);
  hipLaunchKernelGGL(k_build_root, dim3(1), dim3(64), 0, stream, bb, root_keys);

  // grow the tree one level at a time; a level is the run of nodes the previous one created
  std::vector<std::pair<uint32_t, uint32_t>> levels;  // first id, count
  uint32_t big_now = 0, huge_now = 0, wave_now = 0;
  uint32_t first = 0, count = 1, big_seen = 0, huge_seen = 0, wave_seen = 0;
  while (count != 0u) {
    levels.push_back({first, count});
    uint32_t tot[4] = {0, 0, 0, 0};
    if (levels.size() == 1) {
      BUILD_TRY(hipMemcpyAsync(tot, bb.n_tmp, 16, hipMemcpyDeviceToHost, stream));
      BUILD_TRY(hipStreamSynchronize(stream));
    } else {
      tot[1] = big_now;
      tot[2] = huge_now;
      tot[3] = wave_now;
    }
    // the shapes for big nodes only when this level holds a node of their size class
    if (tot[2] != huge_seen) {
      const dim3 cgrid((unsigned)max_chunks), ngrid((unsigned)((max_huge + 255) / 256));
      hipLaunchKernelGGL(k_huge_list, dim3(1), dim3(1024), 0, stream, bb, first, count);
      hipLaunchKernelGGL(k_huge_range, cgrid, dim3(1024), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_count, cgrid, dim3(1024), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_scan, ngrid, dim3(256), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_tables, cgrid, dim3(1024), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_place, cgrid, dim3(1024), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_copy, cgrid, dim3(1024), 0, stream, bb);
      hipLaunchKernelGGL(k_huge_finish, ngrid, dim3(256), 0, stream, bb);
    }
    huge_seen = tot[2];
    if (tot[1] != big_seen) hipLaunchKernelGGL((k_build_split<1024>), dim3(count), dim3(1024), 0, stream, bb, first);
    big_seen = tot[1];
    if (tot[3] != wave_seen) hipLaunchKernelGGL((k_build_split<64>), dim3(count), dim3(64), 0, stream, bb, first);
    wave_seen = tot[3];
    hipLaunchKernelGGL(k_build_split_tiny, dim3((count + 255u) / 256u), dim3(256), 0, stream, bb, first, count);
    BUILD_TRY(hipMemcpyAsync(tot, bb.n_tmp, 16, hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    const uint32_t total = tot[0];
    big_now = tot[1];
    huge_now = tot[2];
    wave_now = tot[3];
    if (total > cap || total < first + count) {
      err = "build: node count out of range";
      return MIRT_ERR_HIP;
    }
    first += count;
    count = total - first;
    if (levels.size() > (size_t)n_faces + 2) {
      err = "build: the tree is deeper than it can be";
      return MIRT_ERR_HIP;
    }
  }
  const uint32_t total = first;
  for (size_t l = levels.size(); l-- > 0;)
    hipLaunchKernelGGL(k_build_up, dim3((levels[l].second + 255u) / 256u), dim3(256), 0, stream, bb, levels[l].first, levels[l].second);
  for (size_t l = 0; l < levels.size(); ++l)
    hipLaunchKernelGGL(k_build_down, dim3((levels[l].second + 255u) / 256u), dim3(256), 0, stream, bb, levels[l].first, levels[l].second, d_out);
  *total_out = total;
  return MIRT_OK;
}

int build_blas_device(hipStream_t stream, const float *positions4, uint32_t n_positions, uint32_t *faces4,
                      uint32_t n_faces, MirtBVHNode *nodes_out, uint32_t nodes_capacity, uint32_t *n_nodes,
                      float *device_ms, void **workspace, size_t *workspace_bytes, std::string &err) {
  BuildBufs bb{};
  bb.nf = n_faces;
  bb.leaf_max = kBlasLeafMax;
  float4 *d_pos = nullptr;
  uint4 *d_faces = nullptr;
  MirtBVHNode *d_out = nullptr;
  unsigned long long *root_keys = nullptr;
  const size_t cap = 2 * (size_t)n_faces;  // a split adds two nodes and there are fewer than n_faces splits
  // a level holds at most n_faces / kHugeNode huge nodes, and their chunks are full but for one per node
  const size_t max_huge = n_faces / kHugeNode + 1, max_chunks = n_faces / kBuildChunk + max_huge + 1;
  // one workspace, kept by the caller between builds and grown when a mesh needs more
  auto carve = [&](unsigned char *base) {
    Carver c{base};
    c.take(&d_pos, n_positions);
    c.take(&d_faces, n_faces);
    c.take(&bb.faces_out, n_faces);
    c.take(&bb.cen, n_faces);
    c.take(&bb.bmn, n_faces);
    c.take(&bb.bmx, n_faces);
    c.take(&bb.perm, n_faces);
    c.take(&bb.perm2, n_faces);
    c.take(&bb.rl, n_faces);
    c.take(&bb.lr, n_faces);
    c.take(&bb.pos_r, n_faces);
    c.take(&bb.pos_lb, n_faces);
    c.take(&bb.tn, cap);
    c.take(&bb.n_tmp, 4);
    c.take(&root_keys, 6);
    c.take(&bb.huge, max_huge);
    c.take(&bb.huge_hdr, 2);
    c.take(&bb.chunk_node, max_chunks);
    c.take(&bb.chunk_r, max_chunks);
    c.take(&bb.chunk_lb, max_chunks);
    c.take(&d_out, cap);
    return c.used;
  };
  const size_t need = carve(nullptr);
  if (*workspace == nullptr || *workspace_bytes < need) {
    if (*workspace) (void)hipFree(*workspace);
    *workspace = nullptr;
    *workspace_bytes = 0;
    BUILD_TRY(hipMalloc(workspace, need));
    *workspace_bytes = need;
  }
  carve(static_cast<unsigned char *>(*workspace));
  bb.pos = d_pos;
  bb.faces = d_faces;
  BUILD_TRY(hipMemcpyAsync(d_pos, positions4, (size_t)n_positions * 16, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemcpyAsync(d_faces, faces4, (size_t)n_faces * 16, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemsetAsync(bb.tn, 0, cap * sizeof(TNode), stream));

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  BUILD_TRY(hipEventCreate(&ev0));
  BUILD_TRY(hipEventCreate(&ev1));
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  } guard{ev0, ev1};
  BUILD_TRY(hipEventRecord(ev0, stream));

  const unsigned fblocks = (n_faces + 255u) / 256u;
  hipLaunchKernelGGL(k_build_prep, dim3(fblocks), dim3(256), 0, stream, bb);
  uint32_t total = 0;
  {
    const int rc = grow_tree(stream, bb, n_faces, cap, max_huge, max_chunks, root_keys, d_out, &total, err);
    if (rc != MIRT_OK) return rc;
  }
  hipLaunchKernelGGL(k_build_faces, dim3(fblocks), dim3(256), 0, stream, bb);
  BUILD_TRY(hipEventRecord(ev1, stream));
  BUILD_TRY(hipGetLastError());

  if (total > nodes_capacity) {
    BUILD_TRY(hipStreamSynchronize(stream));
    *n_nodes = total;
    err = "build_blas: nodes_capacity is too small (n_nodes holds the count needed)";
    return MIRT_ERR_INVALID_ARGUMENT;
  }
  BUILD_TRY(hipMemcpyAsync(nodes_out, d_out, (size_t)total * sizeof(MirtBVHNode), hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipMemcpyAsync(faces4, bb.faces_out, (size_t)n_faces * 16, hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipStreamSynchronize(stream));
  *n_nodes = total;
  if (device_ms) {
    float ms = 0.0f;
    BUILD_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    *device_ms = ms;
  }
  return MIRT_OK;
}

int build_tlas_device(hipStream_t stream, const float *geom_aabbs6, uint32_t n_geometries, const float *mats16,
                      const uint32_t *inst_geometry, uint32_t n_instances, MirtBVHNode *nodes_out, uint32_t nodes_capacity,
                      uint32_t *n_nodes, uint32_t *instance_ids, float *device_ms, void **workspace, size_t *workspace_bytes,
                      std::string &err, const TlasResident *resident) {
  BuildBufs bb{};
  const uint32_t n = n_instances;
  bb.nf = n;
  bb.leaf_max = kTlasLeafMax;
  float *d_aabbs = nullptr, *d_mats = nullptr;
  uint32_t *d_geom = nullptr;
  MirtBVHNode *d_out = nullptr;
  unsigned long long *root_keys = nullptr;
  const size_t cap = 2 * (size_t)n;
  const size_t max_huge = n / kHugeNode + 1, max_chunks = n / kBuildChunk + max_huge + 1;
  auto carve = [&](unsigned char *base) {
    Carver c{base};
    c.take(&d_aabbs, 6 * (size_t)n_geometries);
    c.take(&d_mats, 16 * (size_t)n);
    c.take(&d_geom, n);
    c.take(&bb.cen, n);
    c.take(&bb.bmn, n);
    c.take(&bb.bmx, n);
    c.take(&bb.perm, n);
    c.take(&bb.perm2, n);
    c.take(&bb.rl, n);
    c.take(&bb.lr, n);
    c.take(&bb.pos_r, n);
    c.take(&bb.pos_lb, n);
    c.take(&bb.tn, cap);
    c.take(&bb.n_tmp, 4);
    c.take(&root_keys, 6);
    c.take(&bb.huge, max_huge);
    c.take(&bb.huge_hdr, 2);
    c.take(&bb.chunk_node, max_chunks);
    c.take(&bb.chunk_r, max_chunks);
    c.take(&bb.chunk_lb, max_chunks);
    c.take(&d_out, cap);
    return c.used;
  };
  const size_t need = carve(nullptr);
  if (*workspace == nullptr || *workspace_bytes < need) {
    if (*workspace) (void)hipFree(*workspace);
    *workspace = nullptr;
    *workspace_bytes = 0;
    BUILD_TRY(hipMalloc(workspace, need));
    *workspace_bytes = need;
  }
  carve(static_cast<unsigned char *>(*workspace));
  BUILD_TRY(hipMemcpyAsync(d_aabbs, geom_aabbs6, (size_t)n_geometries * 24, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemcpyAsync(d_mats, mats16, (size_t)n * 64, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemcpyAsync(d_geom, inst_geometry, (size_t)n * 4, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemsetAsync(bb.tn, 0, cap * sizeof(TNode), stream));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  BUILD_TRY(hipEventCreate(&ev0));
  BUILD_TRY(hipEventCreate(&ev1));
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  } guard{ev0, ev1};
  BUILD_TRY(hipEventRecord(ev0, stream));
  hipLaunchKernelGGL(k_tlas_prep, dim3((n + 255u) / 256u), dim3(256), 0, stream, bb, d_aabbs, d_mats, d_geom);
  uint32_t total = 0;
  {
    const int rc = grow_tree(stream, bb, n, cap, max_huge, max_chunks, root_keys, d_out, &total, err);
    if (rc != MIRT_OK) return rc;
  }
  if (resident != nullptr) {
    // stay on the device: pack straight into the scene's TLAS block
    *n_nodes = total;
    if (total > resident->capacity) {
      err = "update_transforms: the TLAS outgrew the room reserved for it";
      return MIRT_ERR_INVALID_ARGUMENT;
    }
    const uint32_t work = total > n ? total : n;
    hipLaunchKernelGGL(k_pack_tlas, dim3((work + 255u) / 256u), dim3(256), 0, stream, d_out, total, bb.perm, n,
                       resident->nodes, resident->big, resident->to_inst);
    BUILD_TRY(hipEventRecord(ev1, stream));
    BUILD_TRY(hipGetLastError());
    if (device_ms) {
      BUILD_TRY(hipStreamSynchronize(stream));
      float ms = 0.0f;
      BUILD_TRY(hipEventElapsedTime(&ms, ev0, ev1));
      *device_ms = ms;
    }
    return MIRT_OK;
  }
  BUILD_TRY(hipEventRecord(ev1, stream));
  BUILD_TRY(hipGetLastError());
  if (total > nodes_capacity) {
    BUILD_TRY(hipStreamSynchronize(stream));
    *n_nodes = total;
    err = "build_tlas: nodes_capacity is too small (n_nodes holds the count needed)";
    return MIRT_ERR_INVALID_ARGUMENT;
  }
  BUILD_TRY(hipMemcpyAsync(nodes_out, d_out, (size_t)total * sizeof(MirtBVHNode), hipMemcpyDeviceToHost, stream));
  // instance_ids: the element at every position after the build's swaps (src/tlas.rs:36-39)
  BUILD_TRY(hipMemcpyAsync(instance_ids, bb.perm, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipStreamSynchronize(stream));
  *n_nodes = total;
  if (device_ms) {
    float ms = 0.0f;
    BUILD_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    *device_ms = ms;
  }
  return MIRT_OK;
}

}  // namespace mirt
