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
//     driven by ONE wave-uniform loop that only walks nodes, so lanes in different
//     instances and at different depths run the same instructions; a lane that hits a
//     leaf waits for the wave's next service point;
//   - there a leaf with kCoopLeaf or more triangles (the builder leaves such leaves
//     behind when its partition puts everything on one side) is tested by the
//     whole wave, 64 triangles per pass, and smaller leaves four lanes to a leaf,
//     sixteen leaves per pass, for the lane that reached it.  A sequential scan
//     accepts every triangle with t2 <= current t, so it ends on the smallest
//     t2 and, among equal t2, on the highest index: the lanes compute exactly
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

// Work distribution: every wave reserves kChunk consecutive queue entries with one
// atomic on the launch's cursor and hands them to its lanes as they free up.
#ifndef MIRT_WORK_CHUNK
#define MIRT_WORK_CHUNK 128u
#endif
constexpr uint32_t kChunk = MIRT_WORK_CHUNK;
// Launches with fewer entries than one chunk per wave of the grid (the per-frame call at 512 x 512 hands every
// lane of the GPU one ray) last as long as their slowest wave: smaller chunks, taken while fewer than 24 lanes
// walk, leave a wave that is stuck with long rays alone and give the others more to do (k_extend per frame at
// 512 x 512, chunk / threshold: 64 / 56 1.050 ms, 64 / 1 1.024, 64 / 16 1.029, 32 / 24 1.022, 32 / 8 1.056,
// 32 / 1 1.263, 16 / 1 1.60).
#ifndef MIRT_SMALL_CHUNK
#define MIRT_SMALL_CHUNK 32u
#endif
#ifndef MIRT_SMALL_REFILL_BELOW
#define MIRT_SMALL_REFILL_BELOW 24
#endif
constexpr uint32_t kSmallChunk = MIRT_SMALL_CHUNK;
constexpr int kSmallRefillBelow = MIRT_SMALL_REFILL_BELOW;
// Launches that fill the grid fetch the rays of 64 entries at a time into registers, ahead of the lanes' need
// (trace_engine's `stage`): a refill then waits for no memory.  Against fetching entry and ray at the refill:
// k_extend -1.3 % on the knot, -2.5 % on the split suzanne, -6 % in the Cornell box (short rays: every service
// point refills), -1 % per frame at 512 x 512.
#ifndef MIRT_STAGE_RAYS
#define MIRT_STAGE_RAYS 1
#endif
#ifndef MIRT_STAGE_MIN_CHUNKS
#define MIRT_STAGE_MIN_CHUNKS 16u  // launches of at least this many kChunk entries per wave of the grid are staged
#endif
constexpr bool kStageRays = MIRT_STAGE_RAYS != 0;
constexpr uint32_t kStageMinChunks = MIRT_STAGE_MIN_CHUNKS;
#ifndef MIRT_TRI36
#define MIRT_TRI36 1
#endif
constexpr bool kTri36 = MIRT_TRI36 != 0;  // lanes fetch their triangles from the packed 36-byte copy
#ifndef MIRT_LONE_LEAF
#define MIRT_LONE_LEAF 1
#endif
constexpr bool kLoneLeaf = MIRT_LONE_LEAF != 0;  // a single waiting lane's small leaf is tested without the list and the pulls
#ifndef MIRT_PREFETCH_HK
#define MIRT_PREFETCH_HK 1
#endif
constexpr bool kPrefetchHk = MIRT_PREFETCH_HK != 0;

struct WorkSource {
  uint32_t *counter;  // zeroed before the launch
  uint32_t count;     // number of queue entries
  const uint32_t *entries;  // the queue: fetch() is handed entries[i] for entry i, or i itself when this is null
  unsigned long long *phase;  // diagnostic builds: sixteen cycle / iteration sums, else unused
  uint32_t *phase_waves;      // diagnostic builds: room for sixteen words per wave of the grid (each wave's own sums), or null
};

#ifdef MIRT_PHASE_TIMING
#define MIRT_PH_DECL unsigned long long ph_t0 = __builtin_amdgcn_s_memtime(), ph_start = ph_t0, ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define MIRT_PH_MARK(k)                                        \
  do {                                                         \
    const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
    ph_acc[k] += ph_now - ph_t0;                               \
    ph_t0 = ph_now;                                            \
  } while (0)
#define MIRT_PH_RESET (ph_t0 = __builtin_amdgcn_s_memtime())
#define MIRT_PH_COUNT(k) (ph_acc[k] += 1ull)
#define MIRT_PH_ADD(k, n) (ph_acc[k] += (unsigned long long)(n))
#define MIRT_PH_NOW() __builtin_amdgcn_s_memtime()
#define MIRT_PH_FLUSH(ptr)                                                        \
  do {                                                                            \
    ph_acc[4] = __builtin_amdgcn_s_memtime() - ph_start;                          \
    if ((ptr) != nullptr && trv_lane() == 0u) {                                   \
      for (int ph_k = 0; ph_k < 16; ++ph_k)                                       \
        if (ph_k != 8) atomicAdd(&(ptr)[ph_k], ph_acc[ph_k]);                     \
      atomicMax(&(ptr)[8], ph_acc[4]); /* the slowest wave */                     \
    }                                                                             \
  } while (0)
#define MIRT_PH_FLUSH_WAVE(ptr)                                                                  \
  do {                                                                                           \
    if ((ptr) != nullptr && trv_lane() == 0u) {                                                  \
      uint32_t *ph_w = (ptr) + 16u * (blockIdx.x * (kEngineBlock / 64u) + (threadIdx.x >> 6));   \
      for (int ph_k = 0; ph_k < 16; ++ph_k) ph_w[ph_k] = (uint32_t)ph_acc[ph_k];                 \
    }                                                                                            \
  } while (0)
#else
#define MIRT_PH_DECL
#define MIRT_PH_MARK(k)
#define MIRT_PH_RESET
#define MIRT_PH_COUNT(k)
#define MIRT_PH_ADD(k, n)
#define MIRT_PH_NOW() 0ull
#define MIRT_PH_FLUSH_WAVE(ptr)
#define MIRT_PH_FLUSH(ptr)
#endif

#ifndef MIRT_ENGINE_BLOCK
#define MIRT_ENGINE_BLOCK 1024u
#endif
// Threads per block of every kernel built on the engine.  One 1024-thread block
// per CU (four waves per SIMD; the kernels use some 107 of the 128 registers per lane that allows): every
// wave of the CU shares one copy of the leaf cache, up to kLeafCacheBytes of LDS.
constexpr uint32_t kEngineBlock = MIRT_ENGINE_BLOCK;
// blocks below 512 threads would need a copy of the leaf cache each and go without
constexpr uint32_t kLeafCacheWords16 = kEngineBlock >= 512u ? kLeafCacheBytes / 16u : 0u;
// resident waves per SIMD the engine kernels are compiled for (bounds their registers)
#ifndef MIRT_ENGINE_WAVES_PER_SIMD
#define MIRT_ENGINE_WAVES_PER_SIMD 4
#endif
#define MIRT_ENGINE_BOUNDS __launch_bounds__(kEngineBlock, MIRT_ENGINE_WAVES_PER_SIMD)
constexpr uint32_t kNoExclude = 0xFFFFFFFFu;

// Per-lane walk flags, kept in one vector register: as lane masks in scalar
// registers every divergent update would cost a three-instruction merge.
enum : uint32_t {
  F_HAS = 1u,     // the lane holds a ray
  F_WALK = 2u,    // ... that is still walking
  F_FOUND = 4u,   // a triangle has been accepted
  F_BLAS = 8u,    // the tree being walked is an instance's BLAS
  F_WRAP = 16u,   // the cursor is back at the root: this tree is done
  F_TDONE = 32u,  // the TLAS cursor has wrapped (instances of its last leaf may remain)
  F_HOLD = 64u    // no step until the service point has dealt with the lane
};

#ifndef MIRT_STEPS_PER_CHECK
#define MIRT_STEPS_PER_CHECK 8
#endif
constexpr int kStepsPerCheck = MIRT_STEPS_PER_CHECK;  // steps between two looks at the service condition

using BufferRsrc = __amdgpu_buffer_rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the value of lane ^ 1 / lane ^ 2 within each group of four lanes (one data-parallel move each)
__device__ __forceinline__ uint32_t quad_xor1(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t quad_xor2(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
}

// the value of the mirror-image lane within each group of eight / sixteen lanes
__device__ __forceinline__ uint32_t row_half_mirror(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t row_mirror(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);
}

// 16 bytes at byte offset `off` of the geometry buffer; offsets past its end read zero.
__device__ __forceinline__ float4 geo_load16(BufferRsrc rsrc, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// One word at a wave-uniform byte offset, read through the scalar cache: it comes back on the scalar counter, so
// a wave does not wait for the vector fetches it has in flight (they return in order) before it may use it.
__device__ __forceinline__ uint32_t geo_sload4(const unsigned char *geo, uint32_t uniform_off) {
  typedef const __attribute__((address_space(4))) uint32_t *ConstWord;
  return *(ConstWord)(geo + (uint32_t)__builtin_amdgcn_readfirstlane((int)uniform_off));
}
// A triangle of the packed copy (DScene::tri36_off): p0, e1, e2, twelve bytes each, as three 12-byte loads (16 + 16 + 4
// bytes at the record's 4-byte alignment are slower than the 48-byte record's three aligned 16-byte loads).
typedef uint32_t uint3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void geo_load_tri36(BufferRsrc rsrc, uint32_t off, v3 &p0, v3 &e1, v3 &e2) {
  const uint3v a = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)off, 0, 0);
  const uint3v b = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)off + 12, 0, 0);
  const uint3v c = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)off + 24, 0, 0);
  p0 = V3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
  e1 = V3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
  e2 = V3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
}
__device__ __forceinline__ uint32_t geo_load4(BufferRsrc rsrc, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
}

// The engine.  `fetch(entry, ray)` loads the ray of a queue entry (WorkSource::entries; called by single lanes); `finish(fin, payload, found, t, best)` is called by
// ALL lanes at every refill point, with fin = true on the lanes whose ray is
// done, so it may use wave-aggregated pushes.
//
// Scheduling.  Lanes of a wave are on nodes and on triangles in a ratio of about five to one: a
// step that served both kinds would run the triangle test for a handful of lanes.  So the step
// loop only walks nodes (one 32-byte fetch, one box test per lane and step), and a lane that
// hits a leaf waits for the service point, entered when a quarter of the walking lanes wait
// (or none can move, or the wave is running low on rays; the condition is looked at every
// kStepsPerCheck steps).  There
//   - leaves of kCoopLeaf triangles or more are scanned by the whole wave, 64 per pass;
//   - the small leaves (the builder's leaves hold up to four triangles, its bail-outs more)
//     are tested four lanes to a leaf, sixteen leaves per pass: each lane takes every fourth
//     triangle with the owner's ray (pulled across lanes), the four reduce to the winner —
//     smallest t2, among equal t2 the highest index, which is what the shader's sequential
//     scan with `t2 <= t` ends on — and the owner pulls it back;
//   - what happens once per instance or per ray follows for all waiting lanes at once: leaving
//     a BLAS, turning a TLAS leaf into a slot range, entering the next instance, ending the
//     walk, handing finished rays over and fetching new ones.
// Nodes of all trees and all triangles live in one buffer addressed by 32-bit byte offsets into
// one buffer descriptor; TLAS and BLAS nodes differ only in the lane's base and ray registers.
// Everything a lane keeps per ray lives in registers; LDS holds the leaf cache and a
// 64-entry list per wave.
//
// STAGED selects how a wave comes by its rays (trace_engine picks by the launch's size): fetched 64 at a time
// ahead of need, or entry and ray fetched when a lane asks.
template <bool ANY, bool COUNT, bool STAGED, class Fetch, class Finish>
__device__ __forceinline__ void engine_loop(const DScene &sc, const WorkSource ws, LaneCounts &lc, Fetch fetch,
                                            Finish finish, uint32_t (&s_own)[kEngineBlock / 64u][64],
                                            uint4 (&s_cache)[kLeafCacheWords16 ? kLeafCacheWords16 : 1u]) {
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = trv_lane();
  const BufferRsrc geo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(sc.geo), 0, (int)sc.geo_bytes, 0x00020000);
  const uint32_t tlas_nbase = sc.tlas_base * 32u;

  // wave-uniform work cursor; small launches take smaller chunks and refill later (kSmallChunk above)
  const bool small_launch = !STAGED && ws.count < kChunk * (kEngineBlock / 64u) * gridDim.x;
  const uint32_t chunk = !small_launch ? kChunk : (kChunk < kSmallChunk ? kChunk : kSmallChunk);
  const int refill_below = small_launch ? kSmallRefillBelow : kRefillBelow;
  uint32_t chunk_next = 0, chunk_end = 0;
  bool exhausted = false;
  // Launches that fill the grid many times over stage their rays: a chunk's rays are fetched by the whole wave (lane l keeps ray l of
  // the chunk in registers) and handed to the lanes that need one by cross-lane reads, and the three dependent round
  // trips behind a ray — work cursor, queue entry, ray record — run one chunk apart: when a chunk's rays are asked
  // for, the entries of the next one and the cursor of the one after it are asked for too, so nothing a refill waits
  // for was requested less than a chunk ago.  (A wave holds up to 64 + 64 + 2 x kChunk entries that way, which smaller
  // launches cannot afford — with 512 entries per wave k_extend lost 11 % to it: those take the direct path below.)
  constexpr bool staged = STAGED;
  RayIn sray;
  sray.pos = sray.dir = V3(0.0f, 0.0f, 0.0f);
  sray.tmax = 0.0f;
  sray.last_inst = sray.last_prim = sray.payload = 0u;
  uint32_t st_next = 0, st_count = 0;  // the staged chunk: rays handed out, rays in it
  uint32_t e_next = 0, n1_count = 0;   // the chunk behind it: its queue entries (one per lane), their number
  uint32_t c_next = 0;                 // the cursor's answer for the reservation behind that (lane 0)
  uint32_t res_base = 0, res_left = 0; // the reservation in use: its next chunk, chunks left in it
  constexpr uint32_t kStageChunk = 64u;
  // one cursor atomic reserves kChunk entries (it is one address for the whole GPU: at 64 entries each a scene
  // of short rays waits for it)
  constexpr uint32_t kStageReserve = kChunk > kStageChunk ? kChunk : kStageChunk;
  auto chunk_entries = [&]() {  // asks for the entries of the next chunk
    if (res_left == 0u) {
      res_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_next);
      res_left = kStageReserve / kStageChunk;
      // past the end of the queue every later cursor is past it too: stop asking
      if (res_base < ws.count && lane == 0u) c_next = atomicAdd(ws.counter, kStageReserve);
    }
    const uint32_t b = res_base < ws.count ? res_base : ws.count;
    const uint32_t e = res_base + kStageChunk < ws.count ? res_base + kStageChunk : ws.count;
    res_base += kStageChunk;
    res_left--;
    n1_count = e - b;
    if (lane < n1_count) e_next = ws.entries != nullptr ? ws.entries[b + lane] : b + lane;
  };
  auto advance = [&]() {  // the staged chunk is used up: the next one moves in
    st_next = 0u;
    st_count = n1_count;
    if (lane < st_count) fetch(e_next, sray);
    if (n1_count != 0u) chunk_entries();
  };
  if (staged) {
    if (lane == 0u) c_next = atomicAdd(ws.counter, kStageReserve);
    chunk_entries();
    exhausted = n1_count == 0u;
  }

  // per-lane walk state
  uint32_t fl = 0u;
  float t = FLT_MAX;
  uint32_t cur = 0;                 // node under the cursor (index within its tree)
  uint32_t nbase = 0, tbase = 0;    // byte offsets of the tree's node 0 and of the instance's triangle 0
  uint32_t excl_prim = kNoExclude;  // the triangle of this instance the ray must not hit, if any
  float tseed = 0.0f;               // what ray_box seeds tmin with: 0 in the TLAS, -FLT_MAX in a BLAS
  // the ray in the space of the tree being walked (world for the TLAS)
  v3 cpos = V3(0.0f, 0.0f, 0.0f), cdir = V3(1.0f, 1.0f, 1.0f), cdinv = V3(1.0f, 1.0f, 1.0f);
  // what is left of the leaf the lane is on: count == 0 means none (BLAS: triangles, TLAS: slots)
  uint32_t leaf_start = 0, leaf_count = 0;
  // touched once per instance or per ray
  v3 wpos = V3(0.0f, 0.0f, 0.0f), wdir = V3(1.0f, 1.0f, 1.0f);  // the world-space ray
  v3 wdinv = V3(1.0f, 1.0f, 1.0f);  // 1 / wdir, kept for the return from an instance (a correctly rounded division is a dozen instructions)
  uint32_t last_inst = 0, last_prim = 0, payload = 0;
  uint32_t tlas_cur = 0, slot_next = 0, slot_stop = 0, cur_inst = 0;  // while in a BLAS: where the TLAS walk stands, the instance
  float best_u = 0.0f, best_v = 0.0f;
  uint32_t best_prim = 0, best_inst = 0;
  uint32_t (&own)[64] = s_own[tid >> 6];
  MIRT_PH_DECL;

  auto accept = [&](float t2, float u, float v, uint32_t p) {
    t = t2;
    best_u = u;
    best_v = v;
    best_prim = p;
    best_inst = cur_inst;
    fl |= F_FOUND;
    if (ANY) fl &= ~F_WALK;  // the first accepted triangle ends an any-hit query
  };

  while (true) {
    const int n_walking = __popcll(__ballot((fl & F_WALK) != 0u));
    const int n_held = __popcll(__ballot((fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD)));
    if ((n_held != 0 && kServiceFrac * n_held >= n_walking) || n_held == n_walking ||
        (!exhausted && n_walking < refill_below)) {
      MIRT_PH_COUNT(0);
#ifdef MIRT_PHASE_TIMING
      const unsigned long long ph_sp0 = MIRT_PH_NOW();
#endif
      // ================================================== service point
      // What the housekeeping below asks memory for first — the instance behind the next TLAS slot of a lane
      // that comes back from a BLAS or stands on a TLAS leaf, and that leaf's size — is requested now, so that it
      // arrives under the leaf tests instead of after them (one round trip less per service point: 1.7 %).
      uint32_t pf_id = 0u, pf_big = 0u;
      if (kPrefetchHk) {
        const bool held = (fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD);
        const bool in_blas = (fl & F_BLAS) != 0u;
        if (held && (in_blas ? ((fl & F_WRAP) != 0u && slot_next < slot_stop) : leaf_count != 0u)) {
          pf_id = sc.tlas_to_inst[in_blas ? slot_next : leaf_start];
          if (!in_blas) pf_big = leaf_count == kBigLeaf ? sc.tlas_big[leaf_start] : leaf_count;
        }
      }
      // ------------------------------------- leaves for the whole wave
      {
        unsigned long long bigmask =
            __ballot((fl & (F_WALK | F_HOLD | F_BLAS)) == (F_WALK | F_HOLD | F_BLAS) && leaf_count >= kCoopLeaf);
        MIRT_PH_RESET;
        while (bigmask != 0ull) {
          const int owner = __ffsll((long long)bigmask) - 1;
          bigmask &= bigmask - 1ull;
          MIRT_PH_COUNT(3);
          // the owner's ray and leaf, wave-uniform
          // (the ray through a lane shuffle, so that it stays in vector registers: an
          // operand read from a scalar register slows every instruction of the scan down)
          auto bcast_f = [&](float x) { return __shfl(x, owner); };
          auto bcast_u = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, owner); };
          const v3 op = V3(bcast_f(cpos.x), bcast_f(cpos.y), bcast_f(cpos.z));
          const v3 od = V3(bcast_f(cdir.x), bcast_f(cdir.y), bcast_f(cdir.z));
          const float ot = bcast_f(t);
          const uint32_t ostart = bcast_u(leaf_start);
          const uint32_t otbase = bcast_u(tbase);
          uint32_t ocount = bcast_u(leaf_count);
          if (ocount == kBigLeaf) ocount = geo_sload4(sc.geo, otbase + ostart * 48u + 12u);  // size stored out of line
          const uint32_t oskip = bcast_u(excl_prim);
          // per-lane best over triangles ostart + lane, + 64, ...
          // key: ANY -> index (first accepted in index order);
          //      else (t2 bits << 32) | ~index (smallest t2, ties to the highest index)
          unsigned long long key = ~0ull;
          float ku = 0.0f, kv = 0.0f, kt = 0.0f;
          // one triangle for this lane; returns true when the lane needs no further ones
          auto consider = [&](uint32_t p, v3 p0, v3 e1, v3 e2) {
            float t2, u, v;
            if (ray_tri_uvt(op, od, p0, e1, e2, t2, u, v) && !(t2 > ot || t2 < 0.0f)) {
              const unsigned long long kk =
                  ANY ? (unsigned long long)p
                      : (((unsigned long long)__float_as_uint(t2 + 0.0f) << 32) | (unsigned long long)(~p));
              if (kk < key) {
                key = kk;
                ku = u;
                kv = v;
                kt = t2;
              }
              return ANY;  // later triangles of this lane have higher indices
            }
            return false;
          };
          const uint32_t ocached = kLeafCacheWords16 != 0u ? geo_sload4(sc.geo, otbase + ostart * 48u + 44u) : 0u;
          if (ocached != 0u && kLeafCacheDirect) {
            // the leaf's scan blocks are in LDS: the same three planes as in memory
            const unsigned char *blk = reinterpret_cast<const unsigned char *>(s_cache) + (ocached - 1u) + lane * 16u;
            for (uint32_t k = lane; k < ocount; k += 64u, blk += kScanBlockBytes) {
              const uint32_t p = ostart + k;
              if (p == oskip) continue;
              const float4 a0 = *reinterpret_cast<const float4 *>(blk), a1 = *reinterpret_cast<const float4 *>(blk + 1024u);
              const float a2 = *reinterpret_cast<const float *>(blk + 2048u - lane * 12u);
              if (consider(p, V3(a0.x, a0.y, a0.z), V3(a0.w, a1.x, a1.y), V3(a1.z, a1.w, a2))) break;
            }
          } else if (ocached != 0u) {
            // the leaf is in LDS: three 16-bit vertex numbers per triangle, e1 and e2 rebuilt
            // with the subtractions the upload does for DTri
            const uint2 *rows = reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned char *>(s_cache) + (ocached - 1u));
            for (uint32_t k = lane; k < ocount; k += 64u) {
              const uint32_t p = ostart + k;
              if (p == oskip) continue;
              const uint2 row = rows[k];
              const uint4 q0 = s_cache[row.x & 0xFFFFu], q1 = s_cache[row.x >> 16], q2 = s_cache[row.y];
              const v3 p0 = V3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
              const v3 p1 = V3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
              const v3 p2 = V3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
              if (consider(p, p0, sub3(p1, p0), sub3(p2, p0))) break;
            }
          } else {
            uint32_t boff = geo_sload4(sc.geo, otbase + ostart * 48u + 28u) + lane * 16u;  // the leaf's scan blocks
            for (uint32_t k = lane; k < ocount; k += 64u, boff += kScanBlockBytes) {
              const uint32_t p = ostart + k;
              if (p == oskip) continue;
              const float4 a0 = geo_load16(geo, boff), a1 = geo_load16(geo, boff + 1024u);
              const float a2 = __uint_as_float(geo_load4(geo, boff + 2048u - lane * 12u));
              if (consider(p, V3(a0.x, a0.y, a0.z), V3(a0.w, a1.x, a1.y), V3(a1.z, a1.w, a2))) break;
            }
          }
          unsigned long long wkey = key;
          for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long other = __shfl_xor(wkey, off);
            wkey = other < wkey ? other : wkey;
          }
          const bool mine = (int)lane == owner;
          if (wkey != ~0ull) {
            const int winner = __ffsll((long long)__ballot(key == wkey)) - 1;
            const float wu = __shfl(ku, winner), wv = __shfl(kv, winner), wt = __shfl(kt, winner);
            const uint32_t wp = ANY ? (uint32_t)wkey : ~(uint32_t)wkey;
            if (mine) accept(wt, wu, wv, wp);
            if (COUNT && mine) {
              // a scan tests every triangle (ANY: up to the first accepted one) except the excluded one
              const uint32_t tested = ANY ? (wp - ostart + 1u) : ocount;
              const bool skipped = oskip >= ostart && oskip < ostart + tested;
              const uint32_t n_tested = tested - (skipped ? 1u : 0u);
              lc.triangles += n_tested;
              if (ocached != 0u) lc.tri_wave_lds += n_tested;
              else lc.tri_wave_mem += n_tested;
            }
          } else if (COUNT && mine) {
            const bool skipped = oskip >= ostart && oskip < ostart + ocount;
            const uint32_t n_tested = ocount - (skipped ? 1u : 0u);
            lc.triangles += n_tested;
            if (ocached != 0u) lc.tri_wave_lds += n_tested;
            else lc.tri_wave_mem += n_tested;
          }
          if (mine) {
            leaf_count = 0u;
            if ((fl & F_WRAP) == 0u) fl &= ~F_HOLD;
          }
        }
        // (1/dir stays in its registers across the scan: the kernel has them to spare at four waves per SIMD, and
        // rebuilding it, as the engine once did to make room, costs three divisions per service point with a scan)
        MIRT_PH_MARK(7);
      }

      // ------------------------------------------ small leaves, four lanes each
      {
        const bool waits = (fl & (F_WALK | F_HOLD | F_BLAS)) == (F_WALK | F_HOLD | F_BLAS) && leaf_count != 0u;
        const unsigned long long smallmask = __ballot(waits);
        if (kLoneLeaf && smallmask != 0ull && (smallmask & (smallmask - 1ull)) == 0ull) {
          // ONE lane waits (a draining wave's usual case, where this sits on the ray's dependent chain): no list, no
          // cross-lane pulls — its ray and leaf go through scalar registers, lane k tests triangle k of the leaf
          // (fewer than kCoopLeaf of them, so one pass), the winner's key is reduced with data-parallel moves inside
          // each row of sixteen lanes and the two rows in use meet in scalar registers.
          const int ol = __ffsll((long long)smallmask) - 1;
          auto rl_u = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, ol); };
          auto rl_f = [&](float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), ol)); };
          const v3 op = V3(rl_f(cpos.x), rl_f(cpos.y), rl_f(cpos.z));
          const v3 od = V3(rl_f(cdir.x), rl_f(cdir.y), rl_f(cdir.z));
          const float ot = rl_f(t);
          const uint32_t ostart = rl_u(leaf_start), ocount = rl_u(leaf_count), oskip = rl_u(excl_prim);
          const uint32_t ofirst = kTri36 ? sc.tri36_off + ((rl_u(tbase) - sc.tri_off) >> 2) * 3u + __umul24(ostart, 36u)
                                         : rl_u(tbase) + __umul24(ostart, 48u);
          static_assert(kCoopLeaf <= 32u, "a small leaf fits the first two rows of sixteen lanes");
          unsigned long long key = ~0ull;
          float ku = 0.0f, kv = 0.0f, kt = 0.0f;
          if (lane < ocount && ostart + lane != oskip) {
            v3 p0, e1, e2;
            if (kTri36) {
              geo_load_tri36(geo, ofirst + __umul24(lane, 36u), p0, e1, e2);
            } else {
              const uint32_t toff = ofirst + __umul24(lane, 48u);
              const float4 a0 = geo_load16(geo, toff), a1 = geo_load16(geo, toff + 16u), a2 = geo_load16(geo, toff + 32u);
              p0 = V3(a0.x, a0.y, a0.z);
              e1 = V3(a1.x, a1.y, a1.z);
              e2 = V3(a2.x, a2.y, a2.z);
            }
            float t2, u, v;
            if (ray_tri_uvt(op, od, p0, e1, e2, t2, u, v) && !(t2 > ot || t2 < 0.0f)) {
              const uint32_t p = ostart + lane;
              key = ANY ? (unsigned long long)p
                        : (((unsigned long long)__float_as_uint(t2 + 0.0f) << 32) | (unsigned long long)(~p));
              ku = u;
              kv = v;
              kt = t2;
            }
          }
          // the smallest key of every row of sixteen lanes, in all of its lanes: lane ^ 1, lane ^ 2, then the mirror
          // image within eight and within sixteen lanes
          unsigned long long rkey = key;
          for (int step = 0; step < 4; ++step) {
            const uint32_t klo = (uint32_t)rkey, khi = (uint32_t)(rkey >> 32);
            uint32_t olo, ohi;
            if (step == 0) { olo = quad_xor1(klo); ohi = quad_xor1(khi); }
            else if (step == 1) { olo = quad_xor2(klo); ohi = quad_xor2(khi); }
            else if (step == 2) { olo = row_half_mirror(klo); ohi = row_half_mirror(khi); }
            else { olo = row_mirror(klo); ohi = row_mirror(khi); }
            const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
            rkey = other < rkey ? other : rkey;
          }
          const unsigned long long r0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rkey >> 32), 0) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rkey, 0);
          const unsigned long long r1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rkey >> 32), 16) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rkey, 16);
          const unsigned long long wkey = r0 < r1 ? r0 : r1;
          const bool won = wkey != ~0ull;
          float wu = 0.0f, wv = 0.0f, wt = 0.0f;
          if (won) {
            const int winner = __ffsll((long long)__ballot(key == wkey)) - 1;
            wu = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ku), winner));
            wv = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(kv), winner));
            wt = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(kt), winner));
          }
          if ((int)lane == ol) {
            const uint32_t wp = ANY ? (uint32_t)wkey : ~(uint32_t)wkey;
            if (COUNT) {
              const uint32_t tested = (ANY && won) ? (wp - leaf_start + 1u) : leaf_count;
              const bool skipped = excl_prim >= leaf_start && excl_prim < leaf_start + tested;
              lc.triangles += tested - (skipped ? 1u : 0u);
            }
            if (won) accept(wt, wu, wv, wp);
            leaf_count = 0u;
            if ((fl & F_WRAP) == 0u) fl &= ~F_HOLD;
          }
        } else if (smallmask != 0ull) {
          const uint32_t n_wait = (uint32_t)__popcll(smallmask);
          const uint32_t rank = trv_rank(smallmask);  // among the waiting lanes, in lane order
          if (waits) own[rank] = lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // byte offset of the leaf's first triangle
          const uint32_t first_off = kTri36 ? sc.tri36_off + ((tbase - sc.tri_off) >> 2) * 3u + __umul24(leaf_start, 36u)
                                            : tbase + __umul24(leaf_start, 48u);
          const uint32_t group = lane >> 2, quad = lane & 3u;
          for (uint32_t round = 0; round * 16u < n_wait; ++round) {
            const uint32_t r = round * 16u + group;
            const bool serve = r < n_wait;
            const int ol = serve ? (int)own[r] : (int)lane;  // the lane this group of four works for
            const v3 op = V3(__shfl(cpos.x, ol), __shfl(cpos.y, ol), __shfl(cpos.z, ol));
            const v3 od = V3(__shfl(cdir.x, ol), __shfl(cdir.y, ol), __shfl(cdir.z, ol));
            const float ot = __shfl(t, ol);
            const uint32_t ofirst = (uint32_t)__shfl((int)first_off, ol);
            const uint32_t ostart = (uint32_t)__shfl((int)leaf_start, ol);
            const uint32_t ocount_all = (uint32_t)__shfl((int)leaf_count, ol);  // pulled by every lane: a lane that sits out must still lend its registers
            const uint32_t ocount = serve ? ocount_all : 0u;
            const uint32_t oskip = (uint32_t)__shfl((int)excl_prim, ol);
            // this lane's best over triangles quad, quad + 4, ... of the leaf
            // key: ANY -> index (first accepted in index order);
            //      else (t2 bits << 32) | ~index (smallest t2, ties to the highest index)
            unsigned long long key = ~0ull;
            float ku = 0.0f, kv = 0.0f, kt = 0.0f;
            for (uint32_t i = quad; __ballot(i < ocount) != 0ull; i += 4u) {
              if (i < ocount && ostart + i != oskip) {
                v3 p0, e1, e2;
                if (kTri36) {
                  geo_load_tri36(geo, ofirst + __umul24(i, 36u), p0, e1, e2);
                } else {
                  const uint32_t toff = ofirst + __umul24(i, 48u);
                  const float4 a0 = geo_load16(geo, toff), a1 = geo_load16(geo, toff + 16u), a2 = geo_load16(geo, toff + 32u);
                  p0 = V3(a0.x, a0.y, a0.z);
                  e1 = V3(a1.x, a1.y, a1.z);
                  e2 = V3(a2.x, a2.y, a2.z);
                }
                float t2, u, v;
                if (ray_tri_uvt(op, od, p0, e1, e2, t2, u, v) && !(t2 > ot || t2 < 0.0f)) {
                  const uint32_t p = ostart + i;
                  const unsigned long long kk =
                      ANY ? (unsigned long long)p
                          : (((unsigned long long)__float_as_uint(t2 + 0.0f) << 32) | (unsigned long long)(~p));
                  if (kk < key) {
                    key = kk;
                    ku = u;
                    kv = v;
                    kt = t2;
                  }
                }
              }
            }
            // the four lanes of a group agree on the winner
            for (int step = 0; step < 2; ++step) {
              const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
              const uint32_t olo = step == 0 ? quad_xor1(klo) : quad_xor2(klo);
              const uint32_t ohi = step == 0 ? quad_xor1(khi) : quad_xor2(khi);
              const uint32_t ou = step == 0 ? quad_xor1(__float_as_uint(ku)) : quad_xor2(__float_as_uint(ku));
              const uint32_t ov = step == 0 ? quad_xor1(__float_as_uint(kv)) : quad_xor2(__float_as_uint(kv));
              const uint32_t otb = step == 0 ? quad_xor1(__float_as_uint(kt)) : quad_xor2(__float_as_uint(kt));
              const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
              if (other < key) {
                key = other;
                ku = __uint_as_float(ou);
                kv = __uint_as_float(ov);
                kt = __uint_as_float(otb);
              }
            }
            // ... and the owner pulls it from the first lane of its group
            const bool mine = waits && rank >= round * 16u && rank < round * 16u + 16u;
            const int from = mine ? (int)((rank - round * 16u) << 2) : (int)lane;
            const uint32_t wlo = (uint32_t)__shfl((int)(uint32_t)key, from), whi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), from);
            const float wu = __shfl(ku, from), wv = __shfl(kv, from), wt = __shfl(kt, from);
            if (mine) {
              const unsigned long long wkey = ((unsigned long long)whi << 32) | wlo;
              const uint32_t wp = ANY ? wlo : ~wlo;
              const bool won = wkey != ~0ull;
              if (COUNT) {
                // every triangle of the leaf is tested (ANY: up to the first accepted one) except the excluded one
                const uint32_t tested = (ANY && won) ? (wp - leaf_start + 1u) : leaf_count;
                const bool skipped = excl_prim >= leaf_start && excl_prim < leaf_start + tested;
                lc.triangles += tested - (skipped ? 1u : 0u);
              }
              if (won) accept(wt, wu, wv, wp);
              leaf_count = 0u;
              if ((fl & F_WRAP) == 0u) fl &= ~F_HOLD;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }

      // ---------------------------------------------------- housekeeping
      // Held lanes with no triangle leaf left: a BLAS walk that is done goes back
      // to the TLAS; a TLAS leaf becomes a slot range; the next instance of the
      // range is entered (shaders/ray_extend.slang:190-203; matrices were built
      // on the host); with nothing left the walk ends.
      {
        const bool hk = (fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD);
        if (__ballot(hk) != 0ull) {
          if (hk) {
            uint32_t slot_i = 0u, slot_end = 0u;
            if ((fl & F_BLAS) != 0u) {
              fl &= ~F_BLAS;
              cpos = wpos;
              cdir = wdir;
              cdinv = wdinv;  // the bits the walk started with
              cur = tlas_cur;
              slot_i = slot_next;
              slot_end = slot_stop;
              nbase = tlas_nbase;
              tseed = 0.0f;
            } else {
              if ((fl & F_WRAP) != 0u) fl |= F_TDONE;
              if (leaf_count != 0u) {
                slot_i = leaf_start;
                slot_end = leaf_start + (kPrefetchHk ? pf_big : (leaf_count == kBigLeaf ? sc.tlas_big[leaf_start] : leaf_count));
                leaf_count = 0u;
              }
            }
            if (slot_i < slot_end) {
              const uint32_t id = kPrefetchHk ? pf_id : sc.tlas_to_inst[slot_i];
              const DInst *inst = &sc.inst[id];
              if (COUNT) lc.instance_visits++;
              const float4 m0 = reinterpret_cast<const float4 *>(inst->mi)[0];
              const float4 m1 = reinterpret_cast<const float4 *>(inst->mi)[1];
              const float4 m2 = reinterpret_cast<const float4 *>(inst->mi)[2];
              tlas_cur = cur;
              slot_next = slot_i + 1u;
              slot_stop = slot_end;
              cur_inst = id;
              nbase = inst->node_base * 32u;
              tbase = sc.tri_off + inst->tri_base * 48u;
              const v3 pos = wpos, dir = wdir;  // the world-space ray
              cpos = V3(__builtin_fmaf(m0.z, pos.z, __builtin_fmaf(m0.y, pos.y, m0.x * pos.x)) + m0.w,
                        __builtin_fmaf(m1.z, pos.z, __builtin_fmaf(m1.y, pos.y, m1.x * pos.x)) + m1.w,
                        __builtin_fmaf(m2.z, pos.z, __builtin_fmaf(m2.y, pos.y, m2.x * pos.x)) + m2.w);
              cdir = V3(__builtin_fmaf(m0.z, dir.z, __builtin_fmaf(m0.y, dir.y, m0.x * dir.x)),
                        __builtin_fmaf(m1.z, dir.z, __builtin_fmaf(m1.y, dir.y, m1.x * dir.x)),
                        __builtin_fmaf(m2.z, dir.z, __builtin_fmaf(m2.y, dir.y, m2.x * dir.x)));
              cdinv = V3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
              excl_prim = id == last_inst ? last_prim : kNoExclude;
              cur = 0u;
              tseed = -FLT_MAX;
              fl = (fl | F_BLAS) & ~(F_WRAP | F_HOLD);
            } else if ((fl & F_TDONE) != 0u) {
              fl &= ~F_WALK;
            } else {
              fl &= ~(F_WRAP | F_HOLD);
            }
          }
        }
      }

      // ---------------------------------------------------------- refill
      {
        const bool fin = (fl & (F_HAS | F_WALK)) == F_HAS;
        if (__ballot(fin) != 0ull) {
          BestHit best;
          best.t = t;
          best.u = best_u;
          best.v = best_v;
          best.prim = best_prim;
          best.inst = best_inst;
          finish(fin, payload, (fl & F_FOUND) != 0u, t, best);
        }
        if ((fl & F_WALK) == 0u) fl = 0u;
        if (!exhausted && __popcll(__ballot((fl & F_WALK) != 0u)) < refill_below) {
          auto install = [&](const RayIn &ray) {
            // a negative tmax marks an entry that traces nothing
            fl = (ray.tmax < 0.0f) ? F_HAS : (F_HAS | F_WALK);
            t = ray.tmax;
            wpos = cpos = ray.pos;
            wdir = cdir = ray.dir;
            cdinv = V3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
            wdinv = cdinv;
            last_inst = ray.last_inst;
            last_prim = ray.last_prim;
            payload = ray.payload;
            cur = 0u;
            nbase = tlas_nbase;
            tseed = 0.0f;
            leaf_count = 0u;
          };
          if constexpr (staged) {
            // hand out staged rays; lanes left without one when the chunk runs out get theirs at the next service
            // point (waiting here for the next chunk's rays costs more than their eight steps)
            const unsigned long long need = __ballot((fl & F_HAS) == 0u);
            if (st_next >= st_count && n1_count != 0u) advance();  // the wave's first refill
            const uint32_t src = st_next + trv_rank(need);
            RayIn ray;
            {
              const int a = (int)((src & 63u) << 2);
              ray.pos.x = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.pos.x)));
              ray.pos.y = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.pos.y)));
              ray.pos.z = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.pos.z)));
              ray.dir.x = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.dir.x)));
              ray.dir.y = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.dir.y)));
              ray.dir.z = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.dir.z)));
              ray.tmax = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(sray.tmax)));
              ray.last_inst = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)sray.last_inst);
              ray.last_prim = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)sray.last_prim);
              ray.payload = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)sray.payload);
            }
            if ((fl & F_HAS) == 0u && src < st_count) install(ray);
            const uint32_t taken = (uint32_t)__popcll(need);
            st_next = st_next + taken < st_count ? st_next + taken : st_count;
            // a chunk that is used up is replaced at once, so that its rays arrive under the steps that follow
            if (st_next >= st_count && n1_count != 0u) advance();
            exhausted = st_next >= st_count && n1_count == 0u;
          } else {
            const unsigned long long need = __ballot((fl & F_HAS) == 0u);
            if (chunk_next >= chunk_end) {
              uint32_t cbase = 0;
              if (lane == 0u) cbase = atomicAdd(ws.counter, chunk);
              cbase = (uint32_t)__shfl((int)cbase, 0);
              chunk_next = cbase < ws.count ? cbase : ws.count;
              chunk_end = cbase + chunk < ws.count ? cbase + chunk : ws.count;
              if (chunk_next >= chunk_end) exhausted = true;
            }
            const uint32_t idx = chunk_next + trv_rank(need);
            if ((fl & F_HAS) == 0u && idx < chunk_end) {
              RayIn ray;
              fetch(ws.entries != nullptr ? ws.entries[idx] : idx, ray);
              install(ray);
            }
            const uint32_t taken = (uint32_t)__popcll(need);
            chunk_next = chunk_next + taken < chunk_end ? chunk_next + taken : chunk_end;
          }
        }
        if (exhausted && __ballot((fl & F_HAS) != 0u) == 0ull) break;
      }
#ifdef MIRT_PHASE_TIMING
      // service-point time, all of it and the part spent with at most 4 / 16 lanes walking
      MIRT_PH_ADD(14, MIRT_PH_NOW() - ph_sp0);
      if (n_walking <= 4) MIRT_PH_ADD(11, MIRT_PH_NOW() - ph_sp0);
      else if (n_walking <= 16) MIRT_PH_ADD(2, MIRT_PH_NOW() - ph_sp0);
#endif
    }
#ifdef MIRT_PHASE_TIMING
    const unsigned long long ph_st0 = MIRT_PH_NOW();
    const int ph_nw = __popcll(__ballot((fl & F_WALK) != 0u));
#endif

    // =============================================================== steps
    // One node for every lane that can move (blasFirstHit / tlasFirstHit,
    // shaders/ray_extend.slang:127-142, 173-188); a lane that hits a leaf, or whose tree is
    // done, waits for the service point.
#pragma unroll 1
    for (int k = 0; k < kStepsPerCheck; ++k) {
      const bool active = (fl & (F_WALK | F_HOLD)) == F_WALK;
      if (__ballot(active) == 0ull) break;
      MIRT_PH_COUNT(5);
      MIRT_PH_ADD(6, __popcll(__ballot((fl & F_WALK) != 0u)));
      MIRT_PH_ADD(1, __popcll(__ballot(active)));
      if (active) {
        const uint32_t off = nbase + (cur << 5);
        const float4 a0 = geo_load16(geo, off), a1 = geo_load16(geo, off + 16u);
        if (COUNT) {
          if ((fl & F_BLAS) != 0u) lc.blas_nodes++;
          else lc.tlas_nodes++;
        }
        const bool hit = ray_box(cpos, cdinv, a0, a1, tseed, t);
        const uint32_t wa = __float_as_uint(a0.w), wb = __float_as_uint(a1.w);
        const uint32_t leaf = wb >> 24;
        const bool leaf_hit = hit && leaf != 0u;
        cur = (hit && leaf == 0u) ? wa : (wb & kSkipMask);
        leaf_start = wa;
        leaf_count = leaf_hit ? leaf : 0u;  // kBigLeaf is resolved by whoever takes the leaf
        const bool wrapped = cur == 0u;
        fl = (fl & ~(F_WRAP | F_HOLD)) | (wrapped ? F_WRAP : 0u) | ((leaf_hit || wrapped) ? F_HOLD : 0u);
      }
#ifdef MIRT_PHASE_TIMING
      if (ph_nw <= 4) MIRT_PH_COUNT(9);
      else if (ph_nw <= 16) MIRT_PH_COUNT(12);
#endif
    }
#ifdef MIRT_PHASE_TIMING
    // step-loop time, all of it and the part spent with at most 4 / 16 lanes walking
    MIRT_PH_ADD(15, MIRT_PH_NOW() - ph_st0);
    if (ph_nw <= 4) MIRT_PH_ADD(10, MIRT_PH_NOW() - ph_st0);
    else if (ph_nw <= 16) MIRT_PH_ADD(13, MIRT_PH_NOW() - ph_st0);
#endif
  }
  MIRT_PH_FLUSH(ws.phase);
  MIRT_PH_FLUSH_WAVE(ws.phase_waves);
}

// Fills the block's leaf cache and runs the engine in the form that suits the launch's size.
template <bool ANY, bool COUNT, class Fetch, class Finish>
__device__ __forceinline__ void trace_engine(const DScene &sc, const WorkSource ws, LaneCounts &lc, Fetch fetch,
                                             Finish finish) {
  __shared__ uint32_t s_own[kEngineBlock / 64u][64];  // per wave: the lanes that wait with a small leaf, in lane order
  __shared__ uint4 s_cache[kLeafCacheWords16 ? kLeafCacheWords16 : 1u];  // the scene's leaf-cache image (layout.h)
  if (kLeafCacheWords16 != 0u) {
    const uint32_t n16 = sc.leaf_cache_bytes / 16u < kLeafCacheWords16 ? sc.leaf_cache_bytes / 16u : kLeafCacheWords16;
    for (uint32_t i = threadIdx.x; i < n16; i += kEngineBlock) s_cache[i] = sc.leaf_cache[i];
    __syncthreads();
  }
  if (kStageRays && ws.count >= kStageMinChunks * kChunk * (kEngineBlock / 64u) * gridDim.x)
    engine_loop<ANY, COUNT, true>(sc, ws, lc, fetch, finish, s_own, s_cache);
  else
    engine_loop<ANY, COUNT, false>(sc, ws, lc, fetch, finish, s_own, s_cache);
}

// The record extensionMain stores for a hit (shaders/ray_extend.slang:86-89
// and 204-207): barycentric position and normal of the winning triangle moved
// to world space with the instance matrix (M for the normal too, not its
// inverse transpose), normal normalised, front_face from the world direction.
__device__ __forceinline__ void finish_hit(const DScene &sc, v3 world_dir, const BestHit &b, v3 &position, v3 &normal,
                                           uint32_t &front_face) {
  const DInst *inst = &sc.inst[b.inst];
  const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri) + 3 * (size_t)(inst->tri_base + b.prim);
#ifdef MIRT_PROBE_NRM_FROM_TRI  // timing probe only (wrong normals): what the second random 48-byte gather costs
  const float4 *nrm4 = tri4;
#else
  const float4 *nrm4 = reinterpret_cast<const float4 *>(sc.trin) + 3 * (size_t)(inst->tri_base + b.prim);
#endif
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
