// walk_pairs.h — the traversal engine as a walk that reads sibling nodes together and keeps a stack.
#pragma once
// included by engine.h, inside namespace mirt

// Per-lane walk flags, kept in one vector register: as lane masks in scalar
// registers every divergent update would cost a three-instruction merge.
enum : uint32_t {
  F_HAS = 1u,     // the lane holds a ray
  F_WALK = 2u,    // ... that is still walking
  F_FOUND = 4u,   // a triangle has been accepted
  F_BLAS = 8u,    // the tree being walked is an instance's BLAS
  F_WRAP = 16u,   // the BLAS under the lane is done (nothing of it is left on the stack)
  F_POP = 32u,    // the next thing to do is to take the top of the stack
  F_HOLD = 64u    // no step until the service point has dealt with the lane
};

#ifndef MIRT_STEPS_PER_CHECK
#define MIRT_STEPS_PER_CHECK 8
#endif
constexpr int kStepsPerCheck = MIRT_STEPS_PER_CHECK;  // steps between two looks at the service condition

constexpr uint32_t kRootCursor = 0xFFFFFFFFu;  // cursor value "the root of the tree": the pair before node 1 whose left half is not a node

using BufferRsrc = __amdgpu_buffer_rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 16 bytes at byte offset `off` of the geometry buffer; offsets past its end read zero.
__device__ __forceinline__ float4 geo_load16(BufferRsrc rsrc, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ uint32_t geo_load4(BufferRsrc rsrc, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
}

// The two halves of rayBoxIntersect's verdict (shaders/ray_extend.slang:94-111) for a box whose
// test is due later, when t may have shrunk.  The shader computes
//   tmin = max(seed, dmin...), tmax = min(t, dmax...), hit = tmin <= tmax
// with min / max that skip a NaN operand.  tmin does not depend on t, and with
//   m = min(dmax...) over the non-NaN ones (NaN if all three are)
// hit is the same as  !(tmin > m)  and  tmin <= t :  the first half is known when the box is
// read, the second is one comparison with the t of the moment the shader would visit the node.
struct BoxLater {
  float tmin;
  bool maybe;  // !(tmin > m)
};
__device__ __forceinline__ void ray_box_pair(v3 pos, v3 dinv, const float4 llo, const float4 lhi, const float4 rlo,
                                             const float4 rhi, float tmin_seed, float t, bool &hit_left, bool &hit_right_now,
                                             BoxLater &right) {
  const bool sx = dinv.x >= 0.0f, sy = dinv.y >= 0.0f, sz = dinv.z >= 0.0f;
  {
    const float dminx = ((sx ? llo.x : lhi.x) - pos.x) * dinv.x, dmaxx = ((sx ? lhi.x : llo.x) - pos.x) * dinv.x;
    const float dminy = ((sy ? llo.y : lhi.y) - pos.y) * dinv.y, dmaxy = ((sy ? lhi.y : llo.y) - pos.y) * dinv.y;
    const float dminz = ((sz ? llo.z : lhi.z) - pos.z) * dinv.z, dmaxz = ((sz ? lhi.z : llo.z) - pos.z) * dinv.z;
    const float tmin = hw_max3(dminz, dminy, hw_max(dminx, tmin_seed));
    const float tmax = hw_min3(dmaxz, dmaxy, hw_min(dmaxx, t));
    hit_left = tmin <= tmax;
  }
  {
    const float dminx = ((sx ? rlo.x : rhi.x) - pos.x) * dinv.x, dmaxx = ((sx ? rhi.x : rlo.x) - pos.x) * dinv.x;
    const float dminy = ((sy ? rlo.y : rhi.y) - pos.y) * dinv.y, dmaxy = ((sy ? rhi.y : rlo.y) - pos.y) * dinv.y;
    const float dminz = ((sz ? rlo.z : rhi.z) - pos.z) * dinv.z, dmaxz = ((sz ? rhi.z : rlo.z) - pos.z) * dinv.z;
    const float tmin = hw_max3(dminz, dminy, hw_max(dminx, tmin_seed));
    const float m = hw_min3(dmaxz, dmaxy, dmaxx);
    right.tmin = tmin;
    right.maybe = !(tmin > m);
    hit_right_now = tmin <= hw_min(m, t);  // a NaN m leaves t: the shader's chain with every dmax skipped
  }
}

// The engine.  `fetch(index, ray)` loads entry `index` of the client's queue
// (called by single lanes); `finish(fin, payload, found, t, best)` is called by
// ALL lanes at every refill point, with fin = true on the lanes whose ray is
// done, so it may use wave-aggregated pushes.
//
// The walk.  The shader's walk is threaded: node -> left child when its box is hit, -> skip
// link otherwise and after a leaf.  Every node it reaches costs one dependent fetch, and every
// inner node that is hit makes it reach BOTH children, one after the other's subtree.  The
// builder stores the two children of a node next to each other (l, l + 1), one 64-byte line, so
// this walk reads them together when their parent is hit, tests both boxes, and keeps the right
// one — when it is not missed outright — on a per-lane stack with its tmin until the left
// subtree is done; then one comparison with the t of that moment decides as the shader would
// (BoxLater).  Half the dependent fetches for the same boxes, tests, order and verdicts: the set
// of triangles tested, their order and every accepted hit are the shader's.
// Stack entries are (a | leaf size << 24, tmin); the first kStackLds of a lane live in LDS, one
// column per thread, deeper ones in the scene's spill rows.
//
// The kernels built on it are bound by the rate of per-lane fetches, then by instruction
// issue, so the loop is arranged to spend as few of both per ray as it can:
//   - one step advances EVERY lane that can move by one unit of work, a pair of sibling nodes
//     (one 64-byte fetch, two box tests), a triangle of a small leaf (one 48-byte fetch, one
//     test) or a pop.  Nodes of all trees and all triangles live in one buffer, so all kinds
//     share the address and fetch instructions (a 32-bit byte offset into one buffer
//     descriptor) and the registers the data lands in; TLAS and BLAS nodes differ only in the
//     lane's base and ray registers;
//   - everything that happens once per instance or per ray is batched at a
//     service point, entered when a quarter of the walking lanes are held (or
//     none can move, or the wave is running low on rays): whole-wave leaves,
//     housekeeping (leaving a BLAS, turning a TLAS leaf into a slot range,
//     entering the next instance), handing over finished rays
//     and fetching new ones.  The condition is looked at every kStepsPerCheck
//     steps.
template <bool ANY, bool COUNT, class Fetch, class Finish>
__device__ __forceinline__ void trace_engine(const DScene &sc, const WorkSource ws, LaneCounts &lc, Fetch fetch,
                                             Finish finish) {
  __shared__ uint32_t s_ref[kStackLds][kEngineBlock];  // stack: a | leaf size << 24
  __shared__ float s_tmin[kStackLds][kEngineBlock];    // stack: the box's tmin
  __shared__ uint4 s_cache[kLeafCacheWords16 ? kLeafCacheWords16 : 1u];  // the scene's leaf-cache image (layout.h)
  const uint32_t tid = threadIdx.x;
  if (kLeafCacheWords16 != 0u) {
    const uint32_t n16 = sc.leaf_cache_bytes / 16u < kLeafCacheWords16 ? sc.leaf_cache_bytes / 16u : kLeafCacheWords16;
    for (uint32_t i = tid; i < n16; i += kEngineBlock) s_cache[i] = sc.leaf_cache[i];
    __syncthreads();
  }
  const uint32_t lane = trv_lane();
  const BufferRsrc geo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(sc.geo), 0, (int)sc.geo_bytes, 0x00020000);
  const BufferRsrc spill = __builtin_amdgcn_make_buffer_rsrc(sc.stack_spill, 0, (int)sc.stack_spill_bytes, 0x00020000);
  const uint32_t spill_row = gridDim.x * kEngineBlock * 8u;              // bytes per spill row: one entry per lane of the grid
  const uint32_t spill_lane = (blockIdx.x * kEngineBlock + tid) * 8u;
  const uint32_t tlas_nbase = sc.tlas_base * 32u;

  // wave-uniform work cursor
  uint32_t chunk_next = 0, chunk_end = 0;
  bool exhausted = false;

  // per-lane walk state
  uint32_t fl = 0u;
  float t = FLT_MAX;
  uint32_t cur = kRootCursor;       // left node of the pair to read next (index within its tree)
  uint32_t nbase = 0, tbase = 0;    // byte offsets of the tree's node 0 and of the instance's triangle 0
  uint32_t excl_prim = kNoExclude;  // the triangle of this instance the ray must not hit, if any
  uint32_t sp = 0, sbase = 0;       // stack height; height at which the tree being walked is done
  // the ray in the space of the tree being walked (world for the TLAS)
  v3 cpos = V3(0.0f, 0.0f, 0.0f), cdir = V3(1.0f, 1.0f, 1.0f), cdinv = V3(1.0f, 1.0f, 1.0f);
  // what is left of the leaf the lane is on: count == 0 means none (BLAS: triangles, TLAS: slots)
  uint32_t leaf_start = 0, leaf_count = 0;
  // touched once per instance or per ray
  v3 wpos = V3(0.0f, 0.0f, 0.0f), wdir = V3(1.0f, 1.0f, 1.0f);  // the world-space ray
  uint32_t last_inst = 0, last_prim = 0, payload = 0;
  uint32_t slot_next = 0, slot_end = 0, cur_inst = 0;  // while in a BLAS: what is left of the TLAS leaf, the instance
  float best_u = 0.0f, best_v = 0.0f;
  uint32_t best_prim = 0, best_inst = 0;
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
  // go to the node a stack entry or a half of a pair describes: word a, leaf size (0 = inner)
  auto take = [&](uint32_t a, uint32_t leaf) {
    if (leaf == 0u) {
      cur = a;
    } else {
      leaf_start = a;
      leaf_count = leaf;  // kBigLeaf is resolved by whoever takes the leaf
      // small leaves are scanned by the lane itself; big ones and every TLAS leaf (all
      // tagged kBigLeaf) wait for the service point
      if (leaf >= kCoopLeaf) fl |= F_HOLD;
    }
  };

  while (true) {
    const int n_walking = __popcll(__ballot((fl & F_WALK) != 0u));
    const int n_held = __popcll(__ballot((fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD)));
    if ((n_held != 0 && kServiceFrac * n_held >= n_walking) || n_held == n_walking ||
        (!exhausted && n_walking < kRefillBelow)) {
      MIRT_PH_COUNT(0);
      // ================================================== service point
      // ------------------------------------- leaves for the whole wave
      {
        unsigned long long bigmask =
            __ballot((fl & (F_WALK | F_HOLD | F_BLAS)) == (F_WALK | F_HOLD | F_BLAS) && leaf_count != 0u);
        const bool any_big = bigmask != 0ull;
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
          if (ocount == kBigLeaf) ocount = geo_load4(geo, otbase + ostart * 48u + 12u);  // size stored out of line
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
          const uint32_t ocached = kLeafCacheWords16 != 0u ? geo_load4(geo, otbase + ostart * 48u + 44u) : 0u;
          if (ocached != 0u) {
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
            uint32_t boff = geo_load4(geo, otbase + ostart * 48u + 28u) + lane * 16u;  // the leaf's scan blocks
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
            // the leaf is done: on with whatever the stack holds
            leaf_count = 0u;
            fl = (fl & ~F_HOLD) | F_POP;
          }
        }
        // 1/dir is not needed by the scan: rebuilding it here (same operation, same
        // bits) instead of keeping it frees its registers for the scan
        if (any_big) cdinv = V3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
        MIRT_PH_MARK(7);
      }

      // ---------------------------------------------------- housekeeping
      // Held lanes with no triangle leaf left: a BLAS walk that is done goes back
      // to the TLAS; a TLAS leaf becomes a slot range; the next instance of the
      // range is entered (shaders/ray_extend.slang:190-203; matrices were built
      // on the host); with the range used up the TLAS walk goes on from its stack.
      {
        const bool hk = (fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD);
        if (__ballot(hk) != 0ull) {
          if (hk) {
            uint32_t slot_i = 0u, slot_e = 0u;
            if ((fl & F_BLAS) != 0u) {
              fl &= ~F_BLAS;
              cpos = wpos;
              cdir = wdir;
              cdinv = V3(1.0f / wdir.x, 1.0f / wdir.y, 1.0f / wdir.z);  // the bits the walk started with
              slot_i = slot_next;
              slot_e = slot_end;
              nbase = tlas_nbase;
              sbase = 0u;
            } else if (leaf_count != 0u) {
              slot_i = leaf_start;
              slot_e = leaf_start + (leaf_count == kBigLeaf ? sc.tlas_big[leaf_start] : leaf_count);
              leaf_count = 0u;
            }
            if (slot_i < slot_e) {
              const uint32_t id = sc.tlas_to_inst[slot_i];
              const DInst *inst = &sc.inst[id];
              if (COUNT) lc.instance_visits++;
              const float4 m0 = reinterpret_cast<const float4 *>(inst->mi)[0];
              const float4 m1 = reinterpret_cast<const float4 *>(inst->mi)[1];
              const float4 m2 = reinterpret_cast<const float4 *>(inst->mi)[2];
              slot_next = slot_i + 1u;
              slot_end = slot_e;
              cur_inst = id;
              nbase = inst->node_base * 32u;
              tbase = sc.tri_off + inst->tri_base * 48u;
              cpos = V3(__builtin_fmaf(m0.z, wpos.z, __builtin_fmaf(m0.y, wpos.y, m0.x * wpos.x)) + m0.w,
                        __builtin_fmaf(m1.z, wpos.z, __builtin_fmaf(m1.y, wpos.y, m1.x * wpos.x)) + m1.w,
                        __builtin_fmaf(m2.z, wpos.z, __builtin_fmaf(m2.y, wpos.y, m2.x * wpos.x)) + m2.w);
              cdir = V3(__builtin_fmaf(m0.z, wdir.z, __builtin_fmaf(m0.y, wdir.y, m0.x * wdir.x)),
                        __builtin_fmaf(m1.z, wdir.z, __builtin_fmaf(m1.y, wdir.y, m1.x * wdir.x)),
                        __builtin_fmaf(m2.z, wdir.z, __builtin_fmaf(m2.y, wdir.y, m2.x * wdir.x)));
              cdinv = V3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
              excl_prim = id == last_inst ? last_prim : kNoExclude;
              cur = kRootCursor;
              sbase = sp;  // what the TLAS walk left on the stack stays below the BLAS walk's entries
              fl = (fl | F_BLAS) & ~(F_WRAP | F_HOLD | F_POP);
            } else {
              // the slots of the TLAS leaf are used up: next, whatever the TLAS walk deferred
              fl = (fl & ~(F_WRAP | F_HOLD)) | F_POP;
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
        if (!exhausted && __popcll(__ballot((fl & F_WALK) != 0u)) < kRefillBelow) {
          const unsigned long long need = __ballot((fl & F_HAS) == 0u);
          if (chunk_next >= chunk_end) {
            uint32_t cbase = 0;
            if (lane == 0u) cbase = atomicAdd(ws.counter, kChunk);
            cbase = (uint32_t)__shfl((int)cbase, 0);
            chunk_next = cbase < ws.count ? cbase : ws.count;
            chunk_end = cbase + kChunk < ws.count ? cbase + kChunk : ws.count;
            if (chunk_next >= chunk_end) exhausted = true;
          }
          const uint32_t idx = chunk_next + trv_rank(need);
          if ((fl & F_HAS) == 0u && idx < chunk_end) {
            RayIn ray;
            fetch(idx, ray);
            // a negative tmax marks an entry that traces nothing
            fl = (ray.tmax < 0.0f) ? F_HAS : (F_HAS | F_WALK);
            t = ray.tmax;
            wpos = cpos = ray.pos;
            wdir = cdir = ray.dir;
            cdinv = V3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
            last_inst = ray.last_inst;
            last_prim = ray.last_prim;
            payload = ray.payload;
            cur = kRootCursor;
            nbase = tlas_nbase;
            sp = 0u;
            sbase = 0u;
            leaf_count = 0u;
          }
          const uint32_t taken = (uint32_t)__popcll(need);
          chunk_next = chunk_next + taken < chunk_end ? chunk_next + taken : chunk_end;
        }
        if (exhausted && __ballot((fl & F_HAS) != 0u) == 0ull) break;
      }
    }

    // =============================================================== steps
    // One unit of work for every lane that can move: the next triangle of its leaf, the pair
    // of nodes under its cursor, or the top of its stack (blasFirstHit / tlasFirstHit,
    // shaders/ray_extend.slang:127-142, 173-188).
#pragma unroll 1
    for (int k = 0; k < kStepsPerCheck; ++k) {
      const bool active = (fl & (F_WALK | F_HOLD)) == F_WALK;
      if (__ballot(active) == 0ull) break;
      MIRT_PH_COUNT(5);
      MIRT_PH_ADD(6, __popcll(__ballot((fl & F_WALK) != 0u)));
      MIRT_PH_ADD(1, __popcll(__ballot(active && leaf_count == 0u)));
      MIRT_PH_ADD(2, __popcll(__ballot(active && leaf_count != 0u)));
      if (active) {
        bool pop = (fl & F_POP) != 0u;
        if (!pop) {
          const bool is_tri = leaf_count != 0u;
          // the root is the right half of the pair that starts one node before the tree
          const uint32_t off = is_tri ? tbase + __umul24(leaf_start, 48u) : nbase + (cur << 5);
          const float4 a0 = geo_load16(geo, off), a1 = geo_load16(geo, off + 16u), a2 = geo_load16(geo, off + 32u);
          float4 a3 = a2;
          if (!is_tri) a3 = geo_load16(geo, off + 48u);  // issued with the other three, before any test
          if (!is_tri) {
            const bool root = cur == kRootCursor;
            const bool blas = (fl & F_BLAS) != 0u;
            // visits as the shader makes them: the left node (the root, for the root's pair) now,
            // the right one when its turn comes — here if the left one is missed, else at the pop
            auto visit = [&]() {
              if (blas) lc.blas_nodes++;
              else lc.tlas_nodes++;
            };
            if (COUNT) visit();
            bool hit_l, hit_r_now;
            BoxLater later;
            // a TLAS box is also rejected when it lies behind the ray: tmin seeded with 0 (ray_box)
            ray_box_pair(cpos, cdinv, a0, a1, a2, a3, blas ? -FLT_MAX : 0.0f, t, hit_l, hit_r_now, later);
            hit_l = hit_l && !root;
            const uint32_t la = __float_as_uint(a0.w), lb = __float_as_uint(a1.w);
            const uint32_t ra = __float_as_uint(a2.w), rb = __float_as_uint(a3.w);
            if (hit_l) {
              // the right child waits for the left subtree, unless it is missed whatever t becomes
              // (a counting build keeps those too, with a NaN tmin no t passes, to count the visit when it is due)
              if (later.maybe || COUNT) {
                const uint32_t ref = (ra & kSkipMask) | (rb & ~kSkipMask);
                const float tm = later.maybe ? later.tmin : __builtin_nanf("");
                if (sp < kStackLds) {
                  s_ref[sp][tid] = ref;
                  s_tmin[sp][tid] = tm;
                } else {
                  u32x2 e;
                  e.x = ref;
                  e.y = __float_as_uint(tm);
                  __builtin_amdgcn_raw_buffer_store_b64(e, spill, (int)((sp - kStackLds) * spill_row + spill_lane), 0, 0);
                }
                sp += 1u;
              }
              take(la & kSkipMask, lb >> 24);
            } else {
              if (COUNT && !root) visit();
              if (hit_r_now) take(ra & kSkipMask, rb >> 24);  // t has not moved since the left box was tested: the shader's verdict
              else pop = true;
            }
          } else {
            const uint32_t p = leaf_start;
            float t2, u, v;
            const bool ok =
                ray_tri_uvt(cpos, cdir, V3(a0.x, a0.y, a0.z), V3(a1.x, a1.y, a1.z), V3(a2.x, a2.y, a2.z), t2, u, v);
            leaf_start = p + 1u;
            leaf_count -= 1u;
            pop = leaf_count == 0u;
            if (p != excl_prim) {
              if (COUNT) lc.triangles++;
              if (ok && !(t2 > t || t2 < 0.0f)) accept(t2, u, v, p);
            }
          }
        }
        if (ANY && (fl & F_WALK) == 0u) pop = false;  // an any-hit query ended with the triangle just accepted
        if (pop) {
          if (sp == sbase) {
            // nothing of this tree is left: a BLAS hands back to the TLAS at the service point,
            // the TLAS walk — and with it the ray — is done
            if ((fl & F_BLAS) != 0u) fl = (fl & ~F_POP) | F_WRAP | F_HOLD;
            else fl &= ~(F_WALK | F_POP);
          } else {
            sp -= 1u;
            uint32_t ref;
            float tm;
            if (sp < kStackLds) {
              ref = s_ref[sp][tid];
              tm = s_tmin[sp][tid];
            } else {
              const u32x2 e = __builtin_amdgcn_raw_buffer_load_b64(spill, (int)((sp - kStackLds) * spill_row + spill_lane), 0, 0);
              ref = e.x;
              tm = __uint_as_float(e.y);
            }
            if (COUNT) {
              if ((fl & F_BLAS) != 0u) lc.blas_nodes++;
              else lc.tlas_nodes++;
            }
            // the second half of the box test, with the t of the moment the shader reaches the node
            if (tm <= t) {
              fl &= ~F_POP;
              take(ref & kSkipMask, ref >> 24);
            } else {
              fl |= F_POP;
            }
          }
        }
      }
    }
  }
  MIRT_PH_FLUSH(ws.phase);
}

