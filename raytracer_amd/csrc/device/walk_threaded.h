// walk_threaded.h — the traversal engine as the shader's threaded (stackless) walk, one node per fetch.
#pragma once
// included by engine.h, inside namespace mirt

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
static_assert(F_HOLD == F_WRAP << 2, "the triangle step derives F_HOLD from F_WRAP with a shift");

#ifndef MIRT_STEPS_PER_CHECK
#define MIRT_STEPS_PER_CHECK 8
#endif
constexpr int kStepsPerCheck = MIRT_STEPS_PER_CHECK;  // steps between two looks at the service condition

using BufferRsrc = __amdgpu_buffer_rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes at byte offset `off` of the geometry buffer; offsets past its end read zero.
__device__ __forceinline__ float4 geo_load16(BufferRsrc rsrc, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ uint32_t geo_load4(BufferRsrc rsrc, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
}

// The engine.  `fetch(index, ray)` loads entry `index` of the client's queue
// (called by single lanes); `finish(fin, payload, found, t, best)` is called by
// ALL lanes at every refill point, with fin = true on the lanes whose ray is
// done, so it may use wave-aggregated pushes.
//
// The kernels built on it are bound by instruction issue, so the loop is
// arranged to spend as few wave instructions per ray as it can:
//   - one step advances EVERY lane that can move by one unit of work, a node
//     (one 32-byte fetch, one box test) or a triangle of a small leaf (one
//     48-byte fetch, one test).  Nodes of all trees and all triangles live in
//     one buffer, so both kinds share the address and fetch instructions (a
//     32-bit byte offset into one buffer descriptor) and the registers the data
//     lands in; TLAS and BLAS nodes differ only in the lane's base and ray
//     registers;
//   - everything that happens once per instance or per ray is batched at a
//     service point, entered when a quarter of the walking lanes are held (or
//     none can move, or the wave is running low on rays): whole-wave leaves,
//     housekeeping (leaving a BLAS, turning a TLAS leaf into a slot range,
//     entering the next instance, ending the walk), handing over finished rays
//     and fetching new ones.  The condition is looked at every kStepsPerCheck
//     steps.
//
// Registers hold only what a step reads (the ray in the current tree's space,
// t, the cursor, the pending leaf); what is touched once per instance or per
// ray — the world-space ray, the TLAS cursor and slot range while a BLAS is
// walked, the instance and exclusion ids, the client's tag and u, v, prim, inst
// of the best hit so far — lives in LDS, one word column per thread.
template <bool ANY, bool COUNT, class Fetch, class Finish>
__device__ __forceinline__ void trace_engine(const DScene &sc, const WorkSource ws, LaneCounts &lc, Fetch fetch,
                                             Finish finish) {
  __shared__ float s_ray[9][kEngineBlock];      // world pos, dir, 1/dir
  __shared__ uint32_t s_tag[3][kEngineBlock];   // last_inst, last_prim, payload
  __shared__ uint32_t s_best[4][kEngineBlock];  // u bits, v bits, prim, inst
  __shared__ uint32_t s_tlas[4][kEngineBlock];  // while in a BLAS: TLAS cursor, next slot, end of the slot range, instance
  __shared__ uint4 s_cache[kLeafCacheWords16 ? kLeafCacheWords16 : 1u];  // the scene's leaf-cache image (layout.h)
  const uint32_t tid = threadIdx.x;
  if (kLeafCacheWords16 != 0u) {
    const uint32_t n16 = sc.leaf_cache_bytes / 16u < kLeafCacheWords16 ? sc.leaf_cache_bytes / 16u : kLeafCacheWords16;
    for (uint32_t i = tid; i < n16; i += kEngineBlock) s_cache[i] = sc.leaf_cache[i];
    __syncthreads();
  }
  const uint32_t lane = trv_lane();
  const BufferRsrc geo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(sc.geo), 0, (int)sc.geo_bytes, 0x00020000);
  const uint32_t tlas_nbase = sc.tlas_base * 32u;

  // wave-uniform work cursor
  uint32_t chunk_next = 0, chunk_end = 0;
  bool exhausted = false;

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
  MIRT_PH_DECL;

  auto accept = [&](float t2, float u, float v, uint32_t p) {
    t = t2;
    s_best[0][tid] = __float_as_uint(u);
    s_best[1][tid] = __float_as_uint(v);
    s_best[2][tid] = p;
    s_best[3][tid] = s_tlas[3][tid];
    fl |= F_FOUND;
    if (ANY) fl &= ~F_WALK;  // the first accepted triangle ends an any-hit query
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
            leaf_count = 0u;
            if ((fl & F_WRAP) == 0u) fl &= ~F_HOLD;
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
      // on the host); with nothing left the walk ends.
      {
        const bool hk = (fl & (F_WALK | F_HOLD)) == (F_WALK | F_HOLD);
        if (__ballot(hk) != 0ull) {
          if (hk) {
            uint32_t slot_i = 0u, slot_end = 0u;
            if ((fl & F_BLAS) != 0u) {
              fl &= ~F_BLAS;
              cpos = V3(s_ray[0][tid], s_ray[1][tid], s_ray[2][tid]);
              cdir = V3(s_ray[3][tid], s_ray[4][tid], s_ray[5][tid]);
              cdinv = V3(s_ray[6][tid], s_ray[7][tid], s_ray[8][tid]);
              cur = s_tlas[0][tid];
              slot_i = s_tlas[1][tid];
              slot_end = s_tlas[2][tid];
              nbase = tlas_nbase;
              tseed = 0.0f;
            } else {
              if ((fl & F_WRAP) != 0u) fl |= F_TDONE;
              if (leaf_count != 0u) {
                slot_i = leaf_start;
                slot_end = leaf_start + (leaf_count == kBigLeaf ? sc.tlas_big[leaf_start] : leaf_count);
                leaf_count = 0u;
              }
            }
            if (slot_i < slot_end) {
              const uint32_t id = sc.tlas_to_inst[slot_i];
              const DInst *inst = &sc.inst[id];
              if (COUNT) lc.instance_visits++;
              const float4 m0 = reinterpret_cast<const float4 *>(inst->mi)[0];
              const float4 m1 = reinterpret_cast<const float4 *>(inst->mi)[1];
              const float4 m2 = reinterpret_cast<const float4 *>(inst->mi)[2];
              s_tlas[0][tid] = cur;
              s_tlas[1][tid] = slot_i + 1u;
              s_tlas[2][tid] = slot_end;
              s_tlas[3][tid] = id;
              nbase = inst->node_base * 32u;
              tbase = sc.tri_off + inst->tri_base * 48u;
              const v3 pos = cpos, dir = cdir;  // the world-space ray
              cpos = V3(__builtin_fmaf(m0.z, pos.z, __builtin_fmaf(m0.y, pos.y, m0.x * pos.x)) + m0.w,
                        __builtin_fmaf(m1.z, pos.z, __builtin_fmaf(m1.y, pos.y, m1.x * pos.x)) + m1.w,
                        __builtin_fmaf(m2.z, pos.z, __builtin_fmaf(m2.y, pos.y, m2.x * pos.x)) + m2.w);
              cdir = V3(__builtin_fmaf(m0.z, dir.z, __builtin_fmaf(m0.y, dir.y, m0.x * dir.x)),
                        __builtin_fmaf(m1.z, dir.z, __builtin_fmaf(m1.y, dir.y, m1.x * dir.x)),
                        __builtin_fmaf(m2.z, dir.z, __builtin_fmaf(m2.y, dir.y, m2.x * dir.x)));
              cdinv = V3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
              excl_prim = id == s_tag[0][tid] ? s_tag[1][tid] : kNoExclude;
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
          best.u = __uint_as_float(s_best[0][tid]);
          best.v = __uint_as_float(s_best[1][tid]);
          best.prim = s_best[2][tid];
          best.inst = s_best[3][tid];
          finish(fin, s_tag[2][tid], (fl & F_FOUND) != 0u, t, best);
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
            cpos = ray.pos;
            cdir = ray.dir;
            cdinv = V3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
            s_ray[0][tid] = cpos.x;
            s_ray[1][tid] = cpos.y;
            s_ray[2][tid] = cpos.z;
            s_ray[3][tid] = cdir.x;
            s_ray[4][tid] = cdir.y;
            s_ray[5][tid] = cdir.z;
            s_ray[6][tid] = cdinv.x;
            s_ray[7][tid] = cdinv.y;
            s_ray[8][tid] = cdinv.z;
            s_tag[0][tid] = ray.last_inst;
            s_tag[1][tid] = ray.last_prim;
            s_tag[2][tid] = ray.payload;
            cur = 0u;
            nbase = tlas_nbase;
            tseed = 0.0f;
            leaf_count = 0u;
          }
          const uint32_t taken = (uint32_t)__popcll(need);
          chunk_next = chunk_next + taken < chunk_end ? chunk_next + taken : chunk_end;
        }
        if (exhausted && __ballot((fl & F_HAS) != 0u) == 0ull) break;
      }
    }

    // =============================================================== steps
    // One unit of work for every lane that can move: the next triangle of its
    // leaf, else the node under its cursor (blasFirstHit / tlasFirstHit,
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
        const bool is_tri = leaf_count != 0u;
        const uint32_t off = is_tri ? tbase + __umul24(leaf_start, 48u) : nbase + (cur << 5);
        const float4 a0 = geo_load16(geo, off), a1 = geo_load16(geo, off + 16u);
        float4 a2 = a1;
        if (is_tri) a2 = geo_load16(geo, off + 32u);  // issued with the other two, before either test
        if (!is_tri) {
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
          // small leaves are scanned by the lane itself; big ones and every TLAS leaf (all
          // tagged kBigLeaf) wait for the service point
          const bool holdc = leaf_hit ? leaf >= kCoopLeaf : wrapped;
          fl = (fl & ~(F_WRAP | F_HOLD)) | (wrapped ? F_WRAP : 0u) | (holdc ? F_HOLD : 0u);
        } else {
          const uint32_t p = leaf_start;
          float t2, u, v;
          const bool ok =
              ray_tri_uvt(cpos, cdir, V3(a0.x, a0.y, a0.z), V3(a1.x, a1.y, a1.z), V3(a2.x, a2.y, a2.z), t2, u, v);
          leaf_start = p + 1u;
          leaf_count -= 1u;
          if (leaf_count == 0u) fl |= (fl & F_WRAP) << 2;  // the tree is done too: F_HOLD
          if (p != excl_prim) {
            if (COUNT) lc.triangles++;
            if (ok && !(t2 > t || t2 < 0.0f)) accept(t2, u, v, p);
          }
        }
      }
    }
  }
  MIRT_PH_FLUSH(ws.phase);
}

