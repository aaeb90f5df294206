// rt_math.h — the arithmetic contract of the device kernels.
//
// Hit ids, radiance sums and pixels are compared bit-for-bit with a CPU
// implementation of the same algorithm, so every floating-point expression
// here has ONE defined evaluation:
//   - f32 only; +,-,*,/ and sqrt correctly rounded (the library is built with
//     -ffp-contract=off and -fhip-fp32-correctly-rounded-divide-sqrt);
//   - fused multiply-adds exist only where this file writes __builtin_fmaf:
//       dot3(a,b)    = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//       cross(a,b).x = fma(a.y,b.z, -(a.z*b.y))   (and cyclic)
//       row·(v,1)    = fma(m2,v.z, fma(m1,v.y, m0*v.x)) + m3
//       row·(v,0)    = fma(m2,v.z, fma(m1,v.y, m0*v.x))
//   - normalize(v) = v * (1 / sqrt(dot3(v,v)));  length(v) = sqrt(dot3(v,v));
//   - min/max return the non-NaN operand;
//   - pow(x,2) = x*x;  pow(x,5) = NaN for x < 0, else (x*x)*(x*x)*x;
//   - everything else is evaluated left to right as written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirt {

struct v3 {
  float x, y, z;
};

__device__ __forceinline__ v3 V3(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(v3 a, v3 b) {
  return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
  return V3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
            __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ float length3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ v3 normalize3(v3 a) {
  float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
  return scale3(a, inv);
}
__device__ __forceinline__ float pow2f(float x) { return x * x; }
__device__ __forceinline__ float pow5f(float x) {
  if (x < 0.0f) return __builtin_nanf("");
  float x2 = x * x;
  return (x2 * x2) * x;
}
__device__ __forceinline__ float fmax_nn(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmin_nn(float a, float b) { return __builtin_fminf(a, b); }

// rows of a 3x4 matrix stored as 12 consecutive floats
__device__ __forceinline__ v3 xform_point(const float *m, v3 p) {
  return V3(__builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, m[0] * p.x)) + m[3],
            __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, m[4] * p.x)) + m[7],
            __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, m[8] * p.x)) + m[11]);
}
__device__ __forceinline__ v3 xform_vector(const float *m, v3 p) {
  return V3(__builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, m[0] * p.x)),
            __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, m[4] * p.x)),
            __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, m[8] * p.x)));
}

// ------------------------------------------------------------------- RNG
// Hybrid Tausworthe generator of shaders/random.slang:9-30.
struct Rng {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ uint32_t taus_step(uint32_t z, int s1, int s2, int s3, uint32_t m) {
  uint32_t b = (((z << s1) ^ z) >> s2);
  return (((z & m) << s3) ^ b);
}
__device__ __forceinline__ float rng_next(Rng &r) {
  r.x = taus_step(r.x, 13, 19, 12, 4294967294u);
  r.y = taus_step(r.y, 2, 25, 4, 4294967288u);
  r.z = taus_step(r.z, 3, 11, 17, 4294967280u);
  r.w = 1664525u * r.w + 1013904223u;
  return 2.3283064365387e-10f * (float)(r.x ^ r.y ^ r.z ^ r.w);
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// Stream keyed by (seed, source index, sample number): the same sample gets
// the same stream whichever path slot, queue position or GPU renders it.
// Every word is kept >= 1000 like the reference's host-side seeds
// (src/pathtracer_state.rs:101-110).
__device__ __forceinline__ Rng rng_init(uint32_t seed, uint32_t source, uint32_t sample_no) {
  uint32_t w[4];
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    uint32_t v = mix32(mix32(mix32(seed + 0x9E3779B9u * (i + 1u)) ^ source) ^ sample_no);
    if (v < 1000u) v += 1000u;
    w[i] = v;
  }
  return Rng{w[0], w[1], w[2], w[3]};
}

}  // namespace mirt
