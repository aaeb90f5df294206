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
// What differs from the shader is scheduling:
//   - the nested do-while loops are flattened into per-lane state machines
//     driven by ONE wave-uniform loop, so lanes in different instances or at
//     different depths run the same instructions;
//   - a lane that reaches a leaf parks it; parked leaves are tested together
//     in a separate phase (node steps and triangle tests are not interleaved
//     at one-lane utilisation);
//   - a leaf with kCoopLeaf or more triangles (the builder leaves such leaves
//     behind when its partition puts everything on one side) is tested by the
//     whole wave, 64 triangles per step, for the lane that reached it.  A scan
//     accepts every triangle with t2 <= current t, so it ends on the smallest
//     t2 and, among equal t2, on the highest index: the wave computes exactly
//     that with a min-reduction on (t2, -index); u, v, t2 of a triangle do not
//     depend on the current t, so the bits match;
//   - rays differ wildly in cost, so a wave does not wait for its slowest ray:
//     when fewer than kRefillBelow lanes are still walking, finished lanes hand
//     their results over and pull new rays (wave-uniform refill point).
#pragma once

#include <float.h>

#include "layout.h"
#include "rt_math.h"

namespace mirt {

#ifndef MIRT_COOP_LEAF
#define MIRT_COOP_LEAF 24u
#endif
#ifndef MIRT_REFILL_BELOW
#define MIRT_REFILL_BELOW 40
#endif
#ifndef MIRT_NODE_BURST
#define MIRT_NODE_BURST 8
#endif
constexpr uint32_t kCoopLeaf = MIRT_COOP_LEAF;
constexpr int kRefillBelow = MIRT_REFILL_BELOW;  // refill when fewer lanes than this are walking
constexpr int kNodeBurst = MIRT_NODE_BURST;      // node-step iterations between leaf phases (upper bound)

struct LaneCounts {
  uint32_t tlas_nodes = 0, instance_visits = 0, blas_nodes = 0, triangles = 0;
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

// rayBoxIntersect (shaders/ray_extend.slang:94-111) with 1/dir hoisted out
// of the walk.  tmin starts at the lowest finite float (float.minValue).
__device__ __forceinline__ bool ray_box(v3 pos, v3 dinv, const float4 lo, const float4 hi, float tmax_in,
                                        float &tmin_out, float &tmax_out) {
  float tmin = -FLT_MAX;
  float tmax = tmax_in;
  {
    const bool s = dinv.x >= 0.0f;
    const float bmin = s ? lo.x : hi.x, bmax = s ? hi.x : lo.x;
    const float dmin = (bmin - pos.x) * dinv.x, dmax = (bmax - pos.x) * dinv.x;
    tmin = fmax_nn(dmin, tmin);
    tmax = fmin_nn(dmax, tmax);
  }
  {
    const bool s = dinv.y >= 0.0f;
    const float bmin = s ? lo.y : hi.y, bmax = s ? hi.y : lo.y;
    const float dmin = (bmin - pos.y) * dinv.y, dmax = (bmax - pos.y) * dinv.y;
    tmin = fmax_nn(dmin, tmin);
    tmax = fmin_nn(dmax, tmax);
  }
  {
    const bool s = dinv.z >= 0.0f;
    const float bmin = s ? lo.z : hi.z, bmax = s ? hi.z : lo.z;
    const float dmin = (bmin - pos.z) * dinv.z, dmax = (bmax - pos.z) * dinv.z;
    tmin = fmax_nn(dmin, tmin);
    tmax = fmin_nn(dmax, tmax);
  }
  tmin_out = tmin;
  tmax_out = tmax;
  return tmin <= tmax;
}

// rayTriIntersect (shaders/ray_extend.slang:53-92) up to the comparison with
// the current t: returns false when the triangle is missed, else t2, u, v.
__device__ __forceinline__ bool ray_tri_uvt(v3 pos, v3 dir, v3 p0, v3 e1, v3 e2, float &t2_out, float &u_out,
                                            float &v_out) {
  const v3 q = cross3(dir, e2);
  const float alpha = dot3(e1, q);
  if (alpha > -(10e-8f) && alpha < 10e-8f) return false;
  const float f = 1.0f / alpha;
  const v3 s = sub3(pos, p0);
  const float u = f * dot3(s, q);
  if (u < 0.0f) return false;
  const v3 r = cross3(s, e1);
  const float v = f * dot3(dir, r);
  if (v < 0.0f || u + v > 1.0f) return false;
  t2_out = f * dot3(e2, r);
  u_out = u;
  v_out = v;
  return true;
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
constexpr uint32_t kChunk = 128u;
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
#define MIRT_PH_COUNT(k) (ph_acc[k] += 1ull)
#define MIRT_PH_FLUSH(ptr)                                                        \
  do {                                                                            \
    ph_acc[4] = __builtin_amdgcn_s_memtime() - ph_start;                          \
    if ((ptr) != nullptr && trv_lane() == 0u)                                     \
      for (int ph_k = 0; ph_k < 8; ++ph_k) atomicAdd(&(ptr)[ph_k], ph_acc[ph_k]); \
  } while (0)
#else
#define MIRT_PH_DECL
#define MIRT_PH_MARK(k)
#define MIRT_PH_COUNT(k)
#define MIRT_PH_FLUSH(ptr)
#endif

// The engine.  `fetch(index, ray)` loads entry `index` of the client's queue
// (called by single lanes); `finish(fin, ray, found, t, best)` is called by ALL
// lanes at every refill point, with fin = true on the lanes whose ray is done,
// so it may use wave-aggregated pushes.
template <bool ANY, bool COUNT, class Fetch, class Finish>
__device__ __forceinline__ void trace_engine(const DScene &sc, const WorkSource ws, LaneCounts &lc, Fetch fetch,
                                             Finish finish) {
  const float4 *tlas4 = reinterpret_cast<const float4 *>(sc.tlas);
  const float4 *blas4 = reinterpret_cast<const float4 *>(sc.blas);
  const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri);
  const uint32_t lane = trv_lane();

  // wave-uniform work cursor
  uint32_t chunk_next = 0, chunk_end = 0;
  bool exhausted = false;

  // per-lane ray and walk state
  RayIn ray;
  ray.pos = V3(0.0f, 0.0f, 0.0f);
  ray.dir = V3(1.0f, 1.0f, 1.0f);
  ray.last_inst = ray.last_prim = 0x7FFFFFFFu;
  ray.tmax = FLT_MAX;
  ray.payload = 0;
  bool has_ray = false, walking = false, found = false;
  float t = FLT_MAX;
  BestHit best;
  best.t = FLT_MAX;
  best.u = best.v = 0.0f;
  best.prim = best.inst = 0;
  v3 dinv = V3(1.0f, 1.0f, 1.0f);
  uint32_t tcur = 0, slot_i = 0, slot_end = 0;
  bool tlas_done = false, in_blas = false, exclude_here = false;
  uint32_t bcur = 0, id = 0, node_base = 0, tri_base = 0;
  v3 rpos = ray.pos, rdir = ray.dir, rdinv = dinv;
  // a leaf waiting for the triangle phase: count == 0 means none
  uint32_t leaf_start = 0, leaf_count = 0;
  MIRT_PH_DECL;

  while (true) {
    MIRT_PH_COUNT(7);
    // ------------------------------------------------------- refill point
    if (__popcll(__ballot(walking)) < kRefillBelow) {
      const bool fin = has_ray && !walking;
      if (__ballot(fin) != 0ull) finish(fin, ray, found, t, best);
      has_ray = has_ray && walking;
      if (!exhausted) {
        const unsigned long long need = __ballot(!has_ray);
        if (chunk_next >= chunk_end) {
          uint32_t base = 0;
          if (lane == 0u) base = atomicAdd(ws.counter, kChunk);
          base = (uint32_t)__shfl((int)base, 0);
          chunk_next = base < ws.count ? base : ws.count;
          chunk_end = base + kChunk < ws.count ? base + kChunk : ws.count;
          if (chunk_next >= chunk_end) exhausted = true;
        }
        const uint32_t idx = chunk_next + trv_rank(need);
        if (!has_ray && idx < chunk_end) {
          fetch(idx, ray);
          has_ray = true;
          walking = !(ray.tmax < 0.0f);  // a negative tmax marks an entry that traces nothing
          found = false;
          t = ray.tmax;
          dinv = V3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
          tcur = 0u;
          tlas_done = false;
          slot_i = slot_end = 0u;
          in_blas = false;
          leaf_count = 0u;
        }
        const uint32_t taken = (uint32_t)__popcll(need);
        chunk_next = chunk_next + taken < chunk_end ? chunk_next + taken : chunk_end;
      }
      if (__ballot(walking) == 0ull) {
        if (exhausted) break;
        continue;  // the chunk ran dry: grab the next one
      }
    }
    MIRT_PH_MARK(0);

    // --------------------------------------------------------- node phase
    // Lanes without a parked leaf advance their walk; the phase ends when no
    // lane can step, after kNodeBurst steps, or when more lanes wait on a
    // parked leaf than are stepping.
    for (int burst = 0; burst < kNodeBurst; ++burst) {
      const bool can_step = walking && leaf_count == 0u;
      const unsigned long long stepping = __ballot(can_step);
      if (stepping == 0ull) break;
      if (burst > 0 && __popcll(stepping) < __popcll(__ballot(walking && leaf_count != 0u))) break;
      MIRT_PH_COUNT(5);
      if (can_step) {
        if (in_blas) {
          // one BLAS node (blasFirstHit, shaders/ray_extend.slang:127-142)
          const size_t ni = (size_t)(node_base + bcur);
          const float4 blo = blas4[2 * ni], bhi = blas4[2 * ni + 1];
          if (COUNT) lc.blas_nodes++;
          float bt0, bt1;
          const bool bhit = ray_box(rpos, rdinv, blo, bhi, t, bt0, bt1);
          const uint32_t ba = __float_as_uint(blo.w), bb = __float_as_uint(bhi.w);
          const uint32_t bleaf = bb >> 24;
          bcur = (bhit && bleaf == 0u) ? ba : (bb & kSkipMask);
          if (bhit && bleaf != 0u) {
            leaf_start = ba;
            leaf_count = bleaf == kBigLeaf ? sc.tri[tri_base + ba].p0w : bleaf;
          }
          if (bcur == 0u) in_blas = false;
        } else if (slot_i < slot_end) {
          // enter the next instance of the current TLAS leaf
          // (shaders/ray_extend.slang:190-203); matrices were built on the host
          id = sc.tlas_to_inst[slot_i++];
          const DInst *inst = &sc.inst[id];
          if (COUNT) lc.instance_visits++;
          const float4 m0 = reinterpret_cast<const float4 *>(inst->mi)[0];
          const float4 m1 = reinterpret_cast<const float4 *>(inst->mi)[1];
          const float4 m2 = reinterpret_cast<const float4 *>(inst->mi)[2];
          node_base = inst->node_base;
          tri_base = inst->tri_base;
          const v3 pos = ray.pos, dir = ray.dir;
          rpos = V3(__builtin_fmaf(m0.z, pos.z, __builtin_fmaf(m0.y, pos.y, m0.x * pos.x)) + m0.w,
                    __builtin_fmaf(m1.z, pos.z, __builtin_fmaf(m1.y, pos.y, m1.x * pos.x)) + m1.w,
                    __builtin_fmaf(m2.z, pos.z, __builtin_fmaf(m2.y, pos.y, m2.x * pos.x)) + m2.w);
          rdir = V3(__builtin_fmaf(m0.z, dir.z, __builtin_fmaf(m0.y, dir.y, m0.x * dir.x)),
                    __builtin_fmaf(m1.z, dir.z, __builtin_fmaf(m1.y, dir.y, m1.x * dir.x)),
                    __builtin_fmaf(m2.z, dir.z, __builtin_fmaf(m2.y, dir.y, m2.x * dir.x)));
          rdinv = V3(1.0f / rdir.x, 1.0f / rdir.y, 1.0f / rdir.z);
          exclude_here = id == ray.last_inst;
          bcur = 0u;
          in_blas = true;
        } else if (!tlas_done) {
          // one TLAS node (tlasFirstHit, shaders/ray_extend.slang:173-188)
          const float4 lo = tlas4[2 * tcur], hi = tlas4[2 * tcur + 1];
          if (COUNT) lc.tlas_nodes++;
          float tmin_aabb, tmax_aabb;
          const bool hit_aabb =
              ray_box(ray.pos, dinv, lo, hi, t, tmin_aabb, tmax_aabb) && (tmax_aabb >= 0.0f || tmin_aabb >= 0.0f);
          const uint32_t wa = __float_as_uint(lo.w), wb = __float_as_uint(hi.w);
          const uint32_t leaf = wb >> 24;
          if (hit_aabb && leaf != 0u) {
            slot_i = wa;
            slot_end = wa + (leaf == kBigLeaf ? sc.tlas_big[tcur] : leaf);
          }
          tcur = (hit_aabb && leaf == 0u) ? wa : (wb & kSkipMask);
          if (tcur == 0u) tlas_done = true;
        } else {
          walking = false;
        }
      }
    }

    MIRT_PH_MARK(1);
    // ------------------------------------------------- small-leaf phase
    // every lane with a parked leaf of fewer than kCoopLeaf triangles scans it
    {
      const bool small = walking && leaf_count != 0u && leaf_count < kCoopLeaf;
      uint32_t j = 0;
      while (__ballot(small && j < leaf_count) != 0ull) {
        if (small && j < leaf_count) {
          const uint32_t p = leaf_start + j;
          if (!(exclude_here && p == ray.last_prim)) {
            const size_t ti = 3 * (size_t)(tri_base + p);
            const float4 a0 = tri4[ti], a1 = tri4[ti + 1], a2 = tri4[ti + 2];
            if (COUNT) lc.triangles++;
            float t2, u, v;
            if (ray_tri_uvt(rpos, rdir, V3(a0.x, a0.y, a0.z), V3(a1.x, a1.y, a1.z), V3(a2.x, a2.y, a2.z), t2, u, v) &&
                !(t2 > t || t2 < 0.0f)) {
              t = t2;
              best.t = t2;
              best.u = u;
              best.v = v;
              best.prim = p;
              best.inst = id;
              found = true;
              if (ANY) {
                walking = false;
                j = leaf_count;  // the first accepted triangle ends an any-hit query
              }
            }
          }
        }
        ++j;
      }
      if (small) leaf_count = 0u;
    }

    MIRT_PH_MARK(2);
    // ------------------------------------- leaves for the whole wave
    {
      unsigned long long bigmask = __ballot(walking && leaf_count >= kCoopLeaf);
      while (bigmask != 0ull) {
        const int owner = __ffsll((long long)bigmask) - 1;
        bigmask &= bigmask - 1ull;
        const v3 op = V3(__shfl(rpos.x, owner), __shfl(rpos.y, owner), __shfl(rpos.z, owner));
        const v3 od = V3(__shfl(rdir.x, owner), __shfl(rdir.y, owner), __shfl(rdir.z, owner));
        const float ot = __shfl(t, owner);
        const uint32_t ostart = (uint32_t)__shfl((int)leaf_start, owner);
        const uint32_t ocount = (uint32_t)__shfl((int)leaf_count, owner);
        const uint32_t otri_base = (uint32_t)__shfl((int)tri_base, owner);
        const uint32_t oskip =
            __shfl((int)exclude_here, owner) ? (uint32_t)__shfl((int)ray.last_prim, owner) : 0xFFFFFFFFu;
        // per-lane best over triangles ostart + lane, + 64, ...
        // key: ANY -> index (first accepted in index order);
        //      else (t2 bits << 32) | ~index (smallest t2, ties to the highest index)
        unsigned long long key = ~0ull;
        float ku = 0.0f, kv = 0.0f, kt = 0.0f;
        for (uint32_t k = lane; k < ocount; k += 64u) {
          MIRT_PH_COUNT(6);
          const uint32_t p = ostart + k;
          if (p == oskip) continue;
          const size_t ti = 3 * (size_t)(otri_base + p);
          const float4 a0 = tri4[ti], a1 = tri4[ti + 1], a2 = tri4[ti + 2];
          float t2, u, v;
          if (ray_tri_uvt(op, od, V3(a0.x, a0.y, a0.z), V3(a1.x, a1.y, a1.z), V3(a2.x, a2.y, a2.z), t2, u, v) &&
              !(t2 > ot || t2 < 0.0f)) {
            const unsigned long long kk =
                ANY ? (unsigned long long)p
                    : (((unsigned long long)__float_as_uint(t2 + 0.0f) << 32) | (unsigned long long)(~p));
            if (kk < key) {
              key = kk;
              ku = u;
              kv = v;
              kt = t2;
            }
            if (ANY) break;  // later triangles of this lane have higher indices
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
          if (mine) {
            t = wt;
            best.t = wt;
            best.u = wu;
            best.v = wv;
            best.prim = wp;
            best.inst = id;
            found = true;
            if (ANY) walking = false;
          }
          if (COUNT && mine) {
            // a scan tests every triangle (ANY: up to the first accepted one) except the excluded one
            const uint32_t tested = ANY ? (wp - ostart + 1u) : ocount;
            const bool skipped = oskip >= ostart && oskip < ostart + tested;
            lc.triangles += tested - (skipped ? 1u : 0u);
          }
        } else if (COUNT && mine) {
          const bool skipped = oskip >= ostart && oskip < ostart + ocount;
          lc.triangles += ocount - (skipped ? 1u : 0u);
        }
        if (mine) leaf_count = 0u;
      }
    }
    MIRT_PH_MARK(3);
  }
  MIRT_PH_FLUSH(ws.phase);
}

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

}  // namespace mirt
