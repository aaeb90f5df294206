// device_util.h — wave64 queue compaction and slot load/store helpers shared
// by the kernel files.
#pragma once

#include "kernels.h"
#include "engine.h"

namespace mirt {

constexpr uint32_t kNoSample = 0xFFFFFFFFu;
constexpr uint32_t kNoPrim = 0x7FFFFFFFu;  // int.maxValue (shaders/sample.slang:83-84)
constexpr float kPi = 3.14159265f;         // float.getPi()

__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t rank_in_mask(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave-aggregated push: one ballot, one atomic per wave per queue, lanes
// ranked with mbcnt.  Must be reached by every lane of the wave.
__device__ __forceinline__ void wave_push(uint32_t *count, uint32_t *data, bool pred, uint32_t value) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const uint32_t n = (uint32_t)__popcll(mask);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)lane_id() == leader) base = atomicAdd(count, n);
  base = (uint32_t)__shfl((int)base, leader);
  if (pred) data[base + rank_in_mask(mask)] = value;
}

// Block-aggregated push to two queues at once for 256-thread blocks: the four
// waves' ballot counts meet in LDS, threads 0 and 1 reserve the two queue
// tails with one atomic each.  Must be reached by every thread of the block.
struct BlockPush2 {
  uint32_t wave_cnt[2][4];
  uint32_t base[2];
};
__device__ __forceinline__ void block_push2(BlockPush2 &sh, uint32_t *count0, uint32_t *data0, bool pred0,
                                            uint32_t *count1, uint32_t *data1, bool pred1, uint32_t value) {
  const uint32_t tid = threadIdx.x, wid = tid >> 6;
  const unsigned long long m0 = __ballot(pred0), m1 = __ballot(pred1);
  if ((tid & 63u) == 0u) {
    sh.wave_cnt[0][wid] = (uint32_t)__popcll(m0);
    sh.wave_cnt[1][wid] = (uint32_t)__popcll(m1);
  }
  __syncthreads();
  if (tid < 2u) {
    const uint32_t total = sh.wave_cnt[tid][0] + sh.wave_cnt[tid][1] + sh.wave_cnt[tid][2] + sh.wave_cnt[tid][3];
    sh.base[tid] = total ? atomicAdd(tid == 0u ? count0 : count1, total) : 0u;
  }
  __syncthreads();
  uint32_t off0 = sh.base[0], off1 = sh.base[1];
  for (uint32_t w = 0; w < wid; ++w) {
    off0 += sh.wave_cnt[0][w];
    off1 += sh.wave_cnt[1][w];
  }
  if (pred0) data0[off0 + rank_in_mask(m0)] = value;
  if (pred1) data1[off1 + rank_in_mask(m1)] = value;
  __syncthreads();
}

__device__ __forceinline__ void load_ray(const DRay *rays, uint32_t i, v3 &pos, v3 &dir, uint32_t &last_inst,
                                         uint32_t &last_prim) {
  const float4 *r4 = reinterpret_cast<const float4 *>(rays + i);
  const float4 a = r4[0], b = r4[1];
  pos = V3(a.x, a.y, a.z);
  dir = V3(a.w, b.x, b.y);
  last_inst = __float_as_uint(b.z);
  last_prim = __float_as_uint(b.w);
}
__device__ __forceinline__ void store_ray(DRay *rays, uint32_t i, v3 pos, v3 dir, uint32_t last_inst, uint32_t last_prim) {
  float4 *r4 = reinterpret_cast<float4 *>(rays + i);
  r4[0] = make_float4(pos.x, pos.y, pos.z, dir.x);
  r4[1] = make_float4(dir.y, dir.z, __uint_as_float(last_inst), __uint_as_float(last_prim));
}

__device__ __forceinline__ Rng load_rng(const DPath *p) {
  const float4 r = reinterpret_cast<const float4 *>(p)[2];
  return Rng{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w)};
}
__device__ __forceinline__ void store_rng(DPath *p, const Rng &rng) {
  reinterpret_cast<float4 *>(p)[2] =
      make_float4(__uint_as_float(rng.x), __uint_as_float(rng.y), __uint_as_float(rng.z), __uint_as_float(rng.w));
}

// One atomic per wave and counter.
__device__ __forceinline__ void flush_counts(unsigned long long *stats, int first, const LaneCounts &lc) {
  const uint32_t v[4] = {lc.tlas_nodes, lc.instance_visits, lc.blas_nodes, lc.triangles};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long s = v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane_id() == 0u && s) atomicAdd(&stats[first + k], s);
  }
}

}  // namespace mirt
