// engine.h — the ray traversal engine: TLAS -> BLAS threaded (stackless) walk
// with ray/box and ray/triangle tests, one ray per lane, scheduled per wave64
// with ray refill.
//
// Arithmetic and visit order are those of tlasFirstHit / blasFirstHit /
// rayBoxIntersect / rayTriIntersect (shaders/ray_extend.slang:53-215): start at
// node 0, go to the left child when the box is hit, to the skip link otherwise
// and after every leaf, stop when the walk returns to node 0; primitives in
// ascending index, a hit with t2 == t replaces the previous one; the TLAS box
// test also rejects boxes entirely behind the ray, the BLAS one does not.
//
// What differs from the shader is scheduling (details at trace_engine):
//   - the nested do-while loops are flattened into per-lane state machines
//     driven by ONE wave-uniform loop, so lanes in different instances, at
//     different depths, on nodes or on triangles run the same instructions;
//   - a leaf with kCoopLeaf or more triangles (the builder leaves such leaves
//     behind when its partition puts everything on one side) is tested by the
//     whole wave, 64 triangles per step, for the lane that reached it.  A scan
//     accepts every triangle with t2 <= current t, so it ends on the smallest
//     t2 and, among equal t2, on the highest index: the wave computes exactly
//     that with a min-reduction on (t2, -index); u, v, t2 of a triangle do not
//     depend on the current t, so the bits match;
//   - rays differ wildly in cost, so a wave does not wait for its slowest ray:
//     waves are persistent, finished lanes hand their results over and pull new
//     rays at wave-uniform service points.
#pragma once

#include <float.h>

#include "layout.h"
#include "rt_math.h"

namespace mirt {

#ifndef MIRT_REFILL_BELOW
#define MIRT_REFILL_BELOW 56
#endif
constexpr uint32_t kCoopLeaf = kScanLeafMin;  // leaves this big go to the whole wave (they have scan blocks)
constexpr int kRefillBelow = MIRT_REFILL_BELOW;  // refill when fewer lanes than this are walking
#ifndef MIRT_SERVICE_FRAC
#define MIRT_SERVICE_FRAC 4
#endif
constexpr int kServiceFrac = MIRT_SERVICE_FRAC;  // service point when 1/kServiceFrac of the walking lanes are held
static_assert(kScanLeafMin <= 255u, "leaves of kBigLeaf triangles or more must go to the whole-wave phase");

struct LaneCounts {
  uint32_t tlas_nodes = 0, instance_visits = 0, blas_nodes = 0, triangles = 0;
  // of `triangles`: those tested in whole-wave scans, by where the scan read them
  // (the LDS leaf cache, or lane-interleaved scan blocks in memory); the rest were
  // fetched one 48-byte record per lane step
  uint32_t tri_wave_lds = 0, tri_wave_mem = 0;
};

struct BestHit {
  float t, u, v;
  uint32_t prim, inst;
};

__device__ __forceinline__ uint32_t trv_lane() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t trv_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// min / max that return the non-NaN operand, as single instructions.  The
// compiler's fminf / fmaxf first quiet a possible signalling NaN with an extra
// v_max per operand; no signalling NaN can reach these (their operands are
// arithmetic results, the seed, and a t that is FLT_MAX, a client tmax checked
// on the host or an accepted t2), so the bare instructions give the same bits.
__device__ __forceinline__ float hw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float hw_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// rayBoxIntersect (shaders/ray_extend.slang:94-111) with 1/dir hoisted out
// of the walk.  The shader starts tmin at the lowest finite float
// (float.minValue) and tlasFirstHit additionally rejects a box with
// tmax < 0 and tmin < 0 (ray_extend.slang:177).  Both are one comparison
// here: with tmin seeded with `tmin_seed` = 0 instead of -FLT_MAX,
//   max(tmin, 0) <= tmax  <=>  tmin <= tmax and 0 <= tmax
// and tmin <= tmax makes (tmax >= 0 or tmin >= 0) the same as tmax >= 0.
// The walk passes 0 in the TLAS and -FLT_MAX in a BLAS.
__device__ __forceinline__ bool ray_box(v3 pos, v3 dinv, const float4 lo, const float4 hi, float tmin_seed,
                                        float tmax_in) {
  const bool sx = dinv.x >= 0.0f, sy = dinv.y >= 0.0f, sz = dinv.z >= 0.0f;
  const float dminx = ((sx ? lo.x : hi.x) - pos.x) * dinv.x, dmaxx = ((sx ? hi.x : lo.x) - pos.x) * dinv.x;
  const float dminy = ((sy ? lo.y : hi.y) - pos.y) * dinv.y, dmaxy = ((sy ? hi.y : lo.y) - pos.y) * dinv.y;
  const float dminz = ((sz ? lo.z : hi.z) - pos.z) * dinv.z, dmaxz = ((sz ? hi.z : lo.z) - pos.z) * dinv.z;
  const float tmin = hw_max3(dminz, dminy, hw_max(dminx, tmin_seed));
  const float tmax = hw_min3(dmaxz, dmaxy, hw_min(dmaxx, tmax_in));
  return tmin <= tmax;
}

// rayTriIntersect (shaders/ray_extend.slang:53-92) up to the comparison with
// the current t: returns false when the triangle is missed, else t2, u, v.
// Written without early exits (the shader's three returns become one
// predicate) so the three 16-byte loads of a triangle are issued together and
// lanes do not diverge; a rejected triangle's u, v, t2 may be inf/NaN and are
// discarded.
__device__ __forceinline__ bool ray_tri_uvt(v3 pos, v3 dir, v3 p0, v3 e1, v3 e2, float &t2_out, float &u_out,
                                            float &v_out) {
  const v3 q = cross3(dir, e2);
  const float alpha = dot3(e1, q);
  const bool flat = alpha > -(10e-8f) && alpha < 10e-8f;
  const float f = 1.0f / alpha;
  const v3 s = sub3(pos, p0);
  const float u = f * dot3(s, q);
  const v3 r = cross3(s, e1);
  const float v = f * dot3(dir, r);
  t2_out = f * dot3(e2, r);
  u_out = u;
  v_out = v;
  return !flat && !(u < 0.0f) && !(v < 0.0f || u + v > 1.0f);
}

// A ray handed to the engine.
struct RayIn {
  v3 pos, dir;
  uint32_t last_inst, last_prim;
  float tmax;        // FLT_MAX for a closest-hit query; negative: the entry traces nothing
  uint32_t payload;  // the client's tag (slot or output index)
};

// Work distribution: every wave grabs chunks of kChunk consecutive queue
// entries with one atomic per chunk and hands them to its lanes as they free up.
#ifndef MIRT_WORK_CHUNK
#define MIRT_WORK_CHUNK 128u
#endif
constexpr uint32_t kChunk = MIRT_WORK_CHUNK;
struct WorkSource {
  uint32_t *counter;  // zeroed before the launch
  uint32_t count;     // number of queue entries
  unsigned long long *phase;  // diagnostic builds: eight cycle / iteration sums, else unused
};

#ifdef MIRT_PHASE_TIMING
#define MIRT_PH_DECL unsigned long long ph_t0 = __builtin_amdgcn_s_memtime(), ph_start = ph_t0, ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MIRT_PH_MARK(k)                                        \
  do {                                                         \
    const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
    ph_acc[k] += ph_now - ph_t0;                               \
    ph_t0 = ph_now;                                            \
  } while (0)
#define MIRT_PH_RESET (ph_t0 = __builtin_amdgcn_s_memtime())
#define MIRT_PH_COUNT(k) (ph_acc[k] += 1ull)
#define MIRT_PH_ADD(k, n) (ph_acc[k] += (unsigned long long)(n))
#define MIRT_PH_FLUSH(ptr)                                                        \
  do {                                                                            \
    ph_acc[4] = __builtin_amdgcn_s_memtime() - ph_start;                          \
    if ((ptr) != nullptr && trv_lane() == 0u)                                     \
      for (int ph_k = 0; ph_k < 8; ++ph_k) atomicAdd(&(ptr)[ph_k], ph_acc[ph_k]); \
  } while (0)
#else
#define MIRT_PH_DECL
#define MIRT_PH_MARK(k)
#define MIRT_PH_RESET
#define MIRT_PH_COUNT(k)
#define MIRT_PH_ADD(k, n)
#define MIRT_PH_FLUSH(ptr)
#endif

#ifndef MIRT_ENGINE_BLOCK
#define MIRT_ENGINE_BLOCK 1024u
#endif
// Threads per block of every kernel built on the engine.  One 1024-thread block
// per CU (four waves per SIMD): the per-thread state takes 80 KB of LDS and the
// leaf cache, which every wave of the CU shares, up to 64 KB more.
constexpr uint32_t kEngineBlock = MIRT_ENGINE_BLOCK;
// the leaf cache needs the one-block-per-CU shape; smaller blocks go without
constexpr uint32_t kLeafCacheWords16 = kEngineBlock == 1024u ? kLeafCacheBytes / 16u : 0u;
constexpr uint32_t kNoExclude = 0xFFFFFFFFu;

// Two walks produce the same hits, visit counts and bits: the threaded one (one node per
// fetch, as the shader walks) and the one that reads sibling nodes together and defers the
// right one on a stack (half the dependent fetches).  Both are bound by instruction issue on
// the benchmark scenes and the threaded one issues fewer scalar instructions, so it is the default.
#ifndef MIRT_WALK_PAIRS
#define MIRT_WALK_PAIRS 0
#endif
#if MIRT_WALK_PAIRS
#include "walk_pairs.h"
#else
#include "walk_threaded.h"
#endif

// The record extensionMain stores for a hit (shaders/ray_extend.slang:86-89
// and 204-207): barycentric position and normal of the winning triangle moved
// to world space with the instance matrix (M for the normal too, not its
// inverse transpose), normal normalised, front_face from the world direction.
__device__ __forceinline__ void finish_hit(const DScene &sc, v3 world_dir, const BestHit &b, v3 &position, v3 &normal,
                                           uint32_t &front_face) {
  const DInst *inst = &sc.inst[b.inst];
  const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri) + 3 * (size_t)(inst->tri_base + b.prim);
  const float4 *nrm4 = reinterpret_cast<const float4 *>(sc.trin) + 3 * (size_t)(inst->tri_base + b.prim);
  const float4 a0 = tri4[0], a1 = tri4[1], a2 = tri4[2];
  const float4 n0 = nrm4[0], n1 = nrm4[1], n2 = nrm4[2];
  const float u = b.u, v = b.v, w = 1.0f - u - v;
  const v3 n = V3(__builtin_fmaf(n2.x, v, __builtin_fmaf(n1.x, u, n0.x * w)),
                  __builtin_fmaf(n2.y, v, __builtin_fmaf(n1.y, u, n0.y * w)),
                  __builtin_fmaf(n2.z, v, __builtin_fmaf(n1.z, u, n0.z * w)));
  const v3 p = V3(__builtin_fmaf(a2.x, v, __builtin_fmaf(a1.x, u, a0.x)),
                  __builtin_fmaf(a2.y, v, __builtin_fmaf(a1.y, u, a0.y)),
                  __builtin_fmaf(a2.z, v, __builtin_fmaf(a1.z, u, a0.z)));
  position = xform_point(inst->m, p);
  normal = normalize3(xform_vector(inst->m, n));
  front_face = dot3(normal, world_dir) < 0.0f ? 1u : 0u;
}

// ---------------------------------------------------------------- textures
// The hit's texture coordinate: the vertex uvs interpolated like the normal
// (shaders/ray_extend.slang:86-88 interpolates position and normal this way;
// the live shader leaves uv at the barycentrics because its host never fills
// Vertex.uv, src/mesh.rs:241).
__device__ __forceinline__ void hit_uv(const DScene &sc, const BestHit &b, float &tu, float &tv) {
  const float4 *q = sc.tri_uv + 2 * (size_t)(sc.inst[b.inst].tri_base + b.prim);
  const float4 a = q[0], c = q[1];
  const float u = b.u, v = b.v, w = 1.0f - u - v;
  tu = __builtin_fmaf(c.x, v, __builtin_fmaf(a.z, u, a.x * w));
  tv = __builtin_fmaf(c.y, v, __builtin_fmaf(a.w, u, a.y * w));
}

__device__ __forceinline__ v3 texel_rgb(uint32_t t) {
  return V3((float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f);
}

// Texture `id` (1-based) at (tu, tv): repeat addressing, nearest or bilinear
// with texel centres at (i + 0.5) / size, channels / 255, every operation in
// the order written here (the CPU restatement writes the same).
__device__ __forceinline__ v3 tex_sample(const DScene &sc, uint32_t id, float tu, float tv) {
  const uint4 t = sc.textures[id - 1u];
  const uint32_t W = t.x, H = t.y;
  const uint32_t *px = sc.texels + t.z;
  const float fu = tu - __builtin_floorf(tu), fv = tv - __builtin_floorf(tv);  // [0, 1); NaN stays NaN
  if (t.w == MIRT_FILTER_NEAREST) {
    // a NaN coordinate converts to texel 0
    uint32_t x = (uint32_t)(fu * (float)W), y = (uint32_t)(fv * (float)H);
    if (!(fu * (float)W >= 0.0f)) x = 0u;
    if (!(fv * (float)H >= 0.0f)) y = 0u;
    if (x >= W) x = W - 1u;
    if (y >= H) y = H - 1u;
    return texel_rgb(px[(size_t)y * W + x]);
  }
  const float sx = fu * (float)W - 0.5f, sy = fv * (float)H - 0.5f;
  const float x0f = __builtin_floorf(sx), y0f = __builtin_floorf(sy);
  const float tx = sx - x0f, ty = sy - y0f;
  // x0f is in [-1, W - 1]; -1 wraps to the last column (NaN: column 0)
  int xi = (x0f >= -1.0f) ? (int)x0f : 0, yi = (y0f >= -1.0f) ? (int)y0f : 0;
  const uint32_t x0 = xi < 0 ? W - 1u : ((uint32_t)xi >= W ? W - 1u : (uint32_t)xi);
  const uint32_t y0 = yi < 0 ? H - 1u : ((uint32_t)yi >= H ? H - 1u : (uint32_t)yi);
  const uint32_t x1 = x0 + 1u == W ? 0u : x0 + 1u, y1 = y0 + 1u == H ? 0u : y0 + 1u;
  const v3 c00 = texel_rgb(px[(size_t)y0 * W + x0]), c10 = texel_rgb(px[(size_t)y0 * W + x1]);
  const v3 c01 = texel_rgb(px[(size_t)y1 * W + x0]), c11 = texel_rgb(px[(size_t)y1 * W + x1]);
  const float ax = 1.0f - tx, ay = 1.0f - ty;
  const v3 top = V3(c00.x * ax + c10.x * tx, c00.y * ax + c10.y * tx, c00.z * ax + c10.z * tx);
  const v3 bot = V3(c01.x * ax + c11.x * tx, c01.y * ax + c11.y * tx, c01.z * ax + c11.z * tx);
  return V3(top.x * ay + bot.x * ty, top.y * ay + bot.y * ty, top.z * ay + bot.z * ty);
}

}  // namespace mirt
