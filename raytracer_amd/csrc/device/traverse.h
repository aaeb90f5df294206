// traverse.h — TLAS -> BLAS threaded (stackless) traversal with ray/box and
// ray/triangle tests.
//
// Same walk as tlasFirstHit / blasFirstHit / rayBoxIntersect /
// rayTriIntersect of shaders/ray_extend.slang:53-215: start at node 0, go to
// the left child when the box is hit, to the skip link otherwise and after
// every leaf, stop when the walk returns to node 0.  Primitives are visited
// in ascending index, a hit with t2 == t replaces the previous one, the TLAS
// box test also rejects boxes entirely behind the ray, the BLAS one does not.
// What differs is only where the operands come from (see layout.h).
#pragma once

#include <float.h>

#include "layout.h"
#include "rt_math.h"

namespace mirt {

struct LaneCounts {
  uint32_t tlas_nodes = 0, instance_visits = 0, blas_nodes = 0, triangles = 0;
};

struct BestHit {
  float t, u, v;
  uint32_t prim, inst;
};

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

// rayTriIntersect (shaders/ray_extend.slang:53-92); the interpolated vertex is
// rebuilt afterwards for the winning triangle only.
__device__ __forceinline__ bool ray_tri(v3 pos, v3 dir, v3 p0, v3 e1, v3 e2, float &t, float &u_out, float &v_out) {
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
  const float t2 = f * dot3(e2, r);
  if (t2 > t || t2 < 0.0f) return false;
  t = t2;
  u_out = u;
  v_out = v;
  return true;
}

// Returns true when something was hit.  ANY: stop at the first accepted
// triangle (connect / shadow query).  COUNT: tally visits into `lc`.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse(const DScene &sc, v3 pos, v3 dir, uint32_t last_inst, uint32_t last_prim,
                                         float &t, BestHit &best, LaneCounts &lc) {
  const v3 dinv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  const float4 *tlas4 = reinterpret_cast<const float4 *>(sc.tlas);
  const float4 *blas4 = reinterpret_cast<const float4 *>(sc.blas);
  const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri);
  bool found = false;
  uint32_t cur = 0;
  do {
    const float4 lo = tlas4[2 * cur], hi = tlas4[2 * cur + 1];
    if (COUNT) lc.tlas_nodes++;
    float tmin_aabb, tmax_aabb;
    const bool hit_aabb = ray_box(pos, dinv, lo, hi, t, tmin_aabb, tmax_aabb) && (tmax_aabb >= 0.0f || tmin_aabb >= 0.0f);
    const uint32_t wa = __float_as_uint(lo.w), wb = __float_as_uint(hi.w);
    const uint32_t leaf = wb >> 24;
    const uint32_t node = cur;
    cur = (hit_aabb && leaf == 0u) ? wa : (wb & kSkipMask);
    if (!hit_aabb || leaf == 0u) continue;

    const uint32_t n_slots = leaf == kBigLeaf ? sc.tlas_big[node] : leaf;
    for (uint32_t i = wa; i < wa + n_slots; ++i) {
      const uint32_t id = sc.tlas_to_inst[i];
      const DInst *inst = &sc.inst[id];
      if (COUNT) lc.instance_visits++;
      const float4 m0 = reinterpret_cast<const float4 *>(inst->mi)[0];
      const float4 m1 = reinterpret_cast<const float4 *>(inst->mi)[1];
      const float4 m2 = reinterpret_cast<const float4 *>(inst->mi)[2];
      const uint32_t node_base = inst->node_base, tri_base = inst->tri_base;
      const v3 rpos = V3(__builtin_fmaf(m0.z, pos.z, __builtin_fmaf(m0.y, pos.y, m0.x * pos.x)) + m0.w,
                         __builtin_fmaf(m1.z, pos.z, __builtin_fmaf(m1.y, pos.y, m1.x * pos.x)) + m1.w,
                         __builtin_fmaf(m2.z, pos.z, __builtin_fmaf(m2.y, pos.y, m2.x * pos.x)) + m2.w);
      const v3 rdir = V3(__builtin_fmaf(m0.z, dir.z, __builtin_fmaf(m0.y, dir.y, m0.x * dir.x)),
                         __builtin_fmaf(m1.z, dir.z, __builtin_fmaf(m1.y, dir.y, m1.x * dir.x)),
                         __builtin_fmaf(m2.z, dir.z, __builtin_fmaf(m2.y, dir.y, m2.x * dir.x)));
      const v3 rdinv = V3(1.0f / rdir.x, 1.0f / rdir.y, 1.0f / rdir.z);
      const bool exclude_here = id == last_inst;

      uint32_t bcur = 0;
      do {
        const float4 blo = blas4[2 * (size_t)(node_base + bcur)], bhi = blas4[2 * (size_t)(node_base + bcur) + 1];
        if (COUNT) lc.blas_nodes++;
        float bt0, bt1;
        const bool bhit = ray_box(rpos, rdinv, blo, bhi, t, bt0, bt1);
        const uint32_t ba = __float_as_uint(blo.w), bb = __float_as_uint(bhi.w);
        const uint32_t bleaf = bb >> 24;
        bcur = (bhit && bleaf == 0u) ? ba : (bb & kSkipMask);
        if (!bhit || bleaf == 0u) continue;

        uint32_t n_tris = bleaf;
        if (bleaf == kBigLeaf) n_tris = sc.tri[tri_base + ba].p0w;
        for (uint32_t p = ba; p < ba + n_tris; ++p) {
          if (exclude_here && p == last_prim) continue;
          const float4 a0 = tri4[3 * (size_t)(tri_base + p)];
          const float4 a1 = tri4[3 * (size_t)(tri_base + p) + 1];
          const float4 a2 = tri4[3 * (size_t)(tri_base + p) + 2];
          if (COUNT) lc.triangles++;
          float u, v;
          if (ray_tri(rpos, rdir, V3(a0.x, a0.y, a0.z), V3(a1.x, a1.y, a1.z), V3(a2.x, a2.y, a2.z), t, u, v)) {
            best.t = t;
            best.u = u;
            best.v = v;
            best.prim = p;
            best.inst = id;
            found = true;
            if (ANY) return true;
          }
        }
      } while (bcur != 0u);
    }
  } while (cur != 0u);
  return found;
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
