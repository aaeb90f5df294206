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

// Block-aggregated push to two queues at once, for blocks of up to 1024
// threads: every wave counts its pushing lanes with one ballot and ranks them
// with mbcnt, the waves' counts meet in LDS, the first wave turns them into
// offsets with a 16-lane scan per queue and reserves the two queue tails with
// one atomic each.  Queue counters are single words that every block of a
// kernel adds to, and same-address atomics are served one after the other
// (about 15 ns apiece on MI355X), so the streaming kernels run 1024-thread
// blocks: one atomic per queue per 1024 entries.  Must be reached by every
// thread of the block.
struct BlockPush2 {
  uint32_t cnt[2][16];
  uint32_t off[2][16];
};
__device__ __forceinline__ void block_push2(BlockPush2 &sh, uint32_t *count0, uint32_t *data0, bool pred0,
                                            uint32_t *count1, uint32_t *data1, bool pred1, uint32_t value) {
  const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
  const uint32_t n_waves = (blockDim.x + 63u) >> 6;
  const unsigned long long m0 = __ballot(pred0), m1 = __ballot(pred1);
  if (lane == 0u) {
    sh.cnt[0][wid] = (uint32_t)__popcll(m0);
    sh.cnt[1][wid] = (uint32_t)__popcll(m1);
  }
  __syncthreads();
  if (wid == 0u) {
    // lanes 0..15: queue 0, lanes 16..31: queue 1; one wave's count per lane
    const uint32_t q = (lane >> 4) & 1u, w = lane & 15u;
    const uint32_t c = (lane < 32u && w < n_waves) ? sh.cnt[q][w] : 0u;
    uint32_t x = c;  // inclusive scan within the 16-lane group
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)x, d, 16);
      if (w >= (uint32_t)d) x += y;
    }
    uint32_t base = 0u;
    if (lane == 15u && x != 0u) base = atomicAdd(count0, x);
    if (lane == 31u && x != 0u) base = atomicAdd(count1, x);
    base = (uint32_t)__shfl((int)base, 15, 16);
    if (lane < 32u && w < n_waves) sh.off[q][w] = base + x - c;
  }
  __syncthreads();
  if (pred0) data0[sh.off[0][wid] + rank_in_mask(m0)] = value;
  if (pred1) data1[sh.off[1][wid] + rank_in_mask(m1)] = value;
  __syncthreads();
}

// Block-aggregated reservation from a 64-bit counter: the lanes with `pred` get
// consecutive numbers, the block's share taken with one atomic.  Must be
// reached by every thread of the block; lanes without `pred` get ~0.
struct BlockReserve {
  uint32_t off[16];
  unsigned long long base;
};
__device__ __forceinline__ unsigned long long block_reserve(BlockReserve &sh, unsigned long long *counter, bool pred) {
  const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
  const uint32_t n_waves = (blockDim.x + 63u) >> 6;
  const unsigned long long m = __ballot(pred);
  if (lane == 0u) sh.off[wid] = (uint32_t)__popcll(m);
  __syncthreads();
  if (wid == 0u) {
    const uint32_t w = lane & 15u;
    const uint32_t c = (lane < 16u && w < n_waves) ? sh.off[w] : 0u;
    uint32_t x = c;  // inclusive scan over the waves' counts
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)x, d, 16);
      if (w >= (uint32_t)d) x += y;
    }
    if (lane == 15u) sh.base = x != 0u ? atomicAdd(counter, (unsigned long long)x) : 0ull;
    if (lane < 16u && w < n_waves) sh.off[w] = x - c;
  }
  __syncthreads();
  const unsigned long long r = pred ? sh.base + sh.off[wid] + rank_in_mask(m) : ~0ull;
  __syncthreads();
  return r;
}

// The same for a thread that pushes to at most one of N queues (`route` in
// [0, N), or negative for none): one atomic per queue per block.
template <int N>
struct BlockPushN {
  uint32_t cnt[N][16];
  uint32_t off[N][16];
};
template <int N>
__device__ __forceinline__ void block_push_n(BlockPushN<N> &sh, uint32_t *const (&count)[N], uint32_t *const (&data)[N],
                                             int route, uint32_t value) {
  const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
  const uint32_t n_waves = (blockDim.x + 63u) >> 6;
  unsigned long long mine = 0ull;  // the ballot of this thread's queue
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const unsigned long long m = __ballot(route == q);
    if (lane == 0u) sh.cnt[q][wid] = (uint32_t)__popcll(m);
    if (route == q) mine = m;
  }
  __syncthreads();
  if (wid == 0u) {
    // four queues at a time: 16 lanes scan the 16 waves' counts of one queue
#pragma unroll
    for (int q0 = 0; q0 < N; q0 += 4) {
      const int q = q0 + (int)(lane >> 4);
      const uint32_t w = lane & 15u;
      const bool live = q < N && w < n_waves;
      const uint32_t c = live ? sh.cnt[q < N ? q : 0][w] : 0u;
      uint32_t x = c;  // inclusive scan within the 16-lane group
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, d, 16);
        if (w >= (uint32_t)d) x += y;
      }
      uint32_t base = 0u;
      if (w == 15u && q < N && x != 0u) base = atomicAdd(count[q < N ? q : 0], x);
      base = (uint32_t)__shfl((int)base, 15, 16);
      if (live) sh.off[q][w] = base + x - c;
    }
  }
  __syncthreads();
  if (route >= 0) data[route][sh.off[route][wid] + rank_in_mask(mine)] = value;
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

// A slot's ray in the path state is laid out differently from the API's MirtRay: position and last instance in
// the first quarter, direction and last triangle in the second —
//   [px, py, pz, last_inst | dx, dy, dz, last_prim]
// — so that the closest-hit walk, which is done with the first quarter once it has fetched the ray, leaves its
// result there with one store (u, v, triangle, instance or kMissInst) and the kernel behind it reads result and
// direction from ONE 32-byte record (there is no separate hit record on that path).
__device__ __forceinline__ void load_pray(const DRay *rays, uint32_t i, v3 &pos, v3 &dir, uint32_t &last_inst,
                                          uint32_t &last_prim) {
  const float4 *r4 = reinterpret_cast<const float4 *>(rays + i);
  const float4 a = r4[0], b = r4[1];
  pos = V3(a.x, a.y, a.z);
  last_inst = __float_as_uint(a.w);
  dir = V3(b.x, b.y, b.z);
  last_prim = __float_as_uint(b.w);
}
__device__ __forceinline__ void store_pray(DRay *rays, uint32_t i, v3 pos, v3 dir, uint32_t last_inst, uint32_t last_prim) {
  float4 *r4 = reinterpret_cast<float4 *>(rays + i);
  r4[0] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(last_inst));
  r4[1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(last_prim));
}

// Quarter k of a slot's path record (DPath: 0 = radiance + bounces, 1 = throughput + source, 2 = stream
// state) and word w of the record (3 = bounces, 7 = source).  kPathPlanes lays the quarters out as three planes
// of `slots` float4 each, so that a wave walking consecutive slots reads a contiguous kilobyte per load.
__device__ __forceinline__ float4 &path_q(const DState &st, uint32_t slot, uint32_t k) {
  return kPathPlanes ? reinterpret_cast<float4 *>(st.path)[(size_t)k * st.slots + slot]
                     : reinterpret_cast<float4 *>(st.path + slot)[k];
}
__device__ __forceinline__ uint32_t &path_w(const DState &st, uint32_t slot, uint32_t w) {
  return reinterpret_cast<uint32_t *>(&path_q(st, slot, w >> 2))[w & 3u];
}
__device__ __forceinline__ Rng load_rng(const DState &st, uint32_t slot) {
  const float4 r = path_q(st, slot, 2u);
  return Rng{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w)};
}
__device__ __forceinline__ void store_rng(const DState &st, uint32_t slot, const Rng &rng) {
  path_q(st, slot, 2u) = make_float4(__uint_as_float(rng.x), __uint_as_float(rng.y), __uint_as_float(rng.z), __uint_as_float(rng.w));
}

// The two whole-wave-scan tallies of a lane, to stats[first], stats[first + 1].
__device__ __forceinline__ void flush_wave_scan_counts(unsigned long long *stats, int first, const LaneCounts &lc) {
  const uint32_t v[2] = {lc.tri_wave_lds, lc.tri_wave_mem};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned long long s = v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane_id() == 0u && s) atomicAdd(&stats[first + k], s);
  }
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
