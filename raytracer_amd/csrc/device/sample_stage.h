// sample_stage.h — accumulateSample and spawnSample (shaders/sample.slang:18-102) as device functions, shared by
// the sample kernel and by the shade kernels that accumulate and respawn the paths that end in them.
#pragma once

#include "device_util.h"

namespace mirt {

// what the closest-hit walk leaves as the instance of a ray that hit nothing
constexpr uint32_t kMissInst = 0xFFFFFFFFu;

// accumulateSample (shaders/sample.slang:35-47).  uint() of a negative or huge
// float is undefined in the reference; this clamps to [0, 2^32).
__device__ __forceinline__ uint32_t to_fixed(float rad, float scale) {
  if (__builtin_isnan(rad) || __builtin_isinf(rad)) rad = 0.0f;
  const float x = rad * scale;
  if (!(x >= 0.0f)) return 0u;
  if (x >= 4294967040.0f) return 4294967040u;
  return (uint32_t)x;
}

// accumulateSample's sums (shaders/sample.slang:38-47) for one finished sample.  The reference's InterlockedAdd
// on 32-bit words wraps; here the sums are 64-bit inside (they cannot wrap: 2^32 adds of the largest value fit)
// and are clamped to 2^32 - 1 wherever they are handed out, so a sum that would pass 2^32 - 1 ends on 2^32 - 1
// whatever the order of the adds — and the adds need not return the old value, which is what they cost: a
// returning atomic's data comes back through the L1's miss queue like a load's.
__device__ __forceinline__ void accumulate_sample(const DState &st, const DFrameParams &fp, uint32_t source, float4 rad) {
  atomicAdd(&st.count[source], 1u);
  unsigned long long *sum = &st.sum3[kSumStride * (size_t)source];
  const uint32_t v0 = to_fixed(rad.x, fp.p.fixed_point_scale), v1 = to_fixed(rad.y, fp.p.fixed_point_scale),
                 v2 = to_fixed(rad.z, fp.p.fixed_point_scale);
  if (v0 != 0u) atomicAdd(sum + 0, (unsigned long long)v0);
  if (v1 != 0u) atomicAdd(sum + 1, (unsigned long long)v1);
  if (v2 != 0u) atomicAdd(sum + 2, (unsigned long long)v2);
  if (st.moments != nullptr) {
    // adaptive sampling: first and second moment of the sample's value y = r + g + b in fixed point,
    // clamped to 24 bits so that 2^16 squares fit 64 bits — integers, so that sums from several
    // GPUs add exactly (the reference declares a float4 sample_std for this, shaders/pathtracer.slang:31,
    // and never writes it)
    unsigned long long y = (unsigned long long)v0 + (unsigned long long)v1 + (unsigned long long)v2;
    if (y > 0xFFFFFFull) y = 0xFFFFFFull;
    atomicAdd(&st.moments[2 * (size_t)source], y);
    atomicAdd(&st.moments[2 * (size_t)source + 1], y * y);
  }
}

// camera basis (shaders/sample.slang:87-90)
struct CamBasis {
  v3 pos, up, right, top_left;
  float dx, dy;
};
__device__ __forceinline__ CamBasis cam_basis(const MirtCamera &cam) {
  CamBasis cb;
  const v3 fwd = V3(cam.forward[0], cam.forward[1], cam.forward[2]);
  cb.pos = V3(cam.position[0], cam.position[1], cam.position[2]);
  cb.up = V3(cam.up[0], cam.up[1], cam.up[2]);
  cb.right = cross3(fwd, cb.up);
  cb.dx = cam.dims[0];
  cb.dy = cam.dims[1];
  cb.top_left = sub3(add3(scale3(fwd, cam.focal_length), scale3(cb.up, cb.dy)), scale3(cb.right, cb.dx));
  return cb;
}

// spawnSample (shaders/sample.slang:59-102): sample number `si` of this render
// call starts in `slot`.
__device__ __forceinline__ void spawn_sample(const DState &st, const DFrameParams &fp, const CamBasis &cb, uint32_t slot,
                                             unsigned long long si) {
  // sample_idx % P (shaders/sample.slang:64-68) over the sources that still take
  // samples: a source flagged Converged is skipped at spawn
  // (shaders/archive/logic.slang:113-116) by not being on the list
  uint32_t source, k;
  if ((si >> 32) == 0ull) {
    source = (uint32_t)si % st.n_active;
    k = (uint32_t)si / st.n_active;
  } else {
    source = (uint32_t)(si % st.n_active);
    k = (uint32_t)(si / st.n_active);
  }
  if (st.active != nullptr) source = st.active[source];
  k += fp.first_sample;
  const MirtSampleSource src = st.sources[source];
  // per-sample streams are keyed by (seed, source, sample number); per-slot streams are the
  // slot's own, seeded by the host and never re-seeded (shaders/random.slang:32-36)
  Rng rng = fp.p.rng_mode == MIRT_RNG_PER_SLOT ? load_rng(st, slot) : rng_init(fp.p.seed, source, k);
  const float r1 = rng_next(rng);  // x jitter takes the first draw
  const float r2 = rng_next(rng);
  const float jx = r1 / (float)st.width, jy = r2 / (float)st.height;
  const float sy = src.screen_pos[1] + jy, sx = src.screen_pos[0] + jx;
  const v3 offset =
      add3(scale3(scale3(scale3(cb.up, -2.0f), cb.dy), sy), scale3(scale3(scale3(cb.right, 2.0f), cb.dx), sx));
  const v3 dir = normalize3(add3(cb.top_left, offset));
  path_q(st, slot, 0u) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(fp.p.max_bounces));
  path_q(st, slot, 1u) = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(source));
  store_rng(st, slot, rng);
  store_pray(st.ray, slot, cb.pos, dir, kNoPrim, kNoPrim);
}

}  // namespace mirt
