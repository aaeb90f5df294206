// gather_bench.hip — how fast can gfx950 chase per-lane dependent pointers?
// Every lane walks its own chain through a table of 32-byte records (the
// access pattern of a BVH walk: one small random record per lane per step,
// the next address taken from the record).  Reports lane-steps per second for
// a sweep of table sizes, resident waves per SIMD, active lanes per wave and
// record fetch widths.  Diagnostic only; not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o build/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                             \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

// WIDTH = bytes fetched per step (16, 32, 64); the chain link is the first word.
template <int WIDTH>
__global__ __launch_bounds__(256) void k_chase(const uint4 *__restrict__ table, uint32_t n_records, uint32_t steps,
                                               uint32_t active_lanes, uint32_t *__restrict__ out) {
  extern __shared__ uint32_t pad[];  // only to limit occupancy
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t idx = (gid * 2654435761u) % n_records;
  uint32_t acc = 0;
  if (lane < active_lanes) {
    for (uint32_t s = 0; s < steps; ++s) {
      const uint4 *p = table + (size_t)idx * 2u;
      uint4 a = p[0];
      acc += a.y;
      if (WIDTH >= 32) {
        uint4 b = p[1];
        acc += b.x ^ b.w;
      }
      if (WIDTH >= 64) {
        uint4 c = p[2], d = p[3];
        acc += c.x ^ d.w;
      }
      idx = a.x;
    }
  }
  if (acc == 0x12345678u) out[gid] = acc + pad[0];
}

// Two independent chains per lane (memory-level parallelism inside a lane).
__global__ __launch_bounds__(256) void k_chase2(const uint4 *__restrict__ table, uint32_t n_records, uint32_t steps,
                                                uint32_t active_lanes, uint32_t *__restrict__ out) {
  extern __shared__ uint32_t pad[];
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t i0 = (gid * 2654435761u) % n_records, i1 = (gid * 40503u + 77u) % n_records;
  uint32_t acc = 0;
  if (lane < active_lanes) {
    for (uint32_t s = 0; s < steps; ++s) {
      const uint4 *p = table + (size_t)i0 * 2u, *q = table + (size_t)i1 * 2u;
      uint4 a = p[0], b = p[1], c = q[0], d = q[1];
      acc += a.y ^ b.w ^ c.y ^ d.w;
      i0 = a.x;
      i1 = c.x;
    }
  }
  if (acc == 0x12345678u) out[gid] = acc + pad[0];
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.gcnArchName, cus);
  uint32_t *d_out;
  CK(hipMalloc(&d_out, 1u << 24));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint32_t steps = 400;
  const size_t sizes_mb[] = {2, 20, 64, 512};
  for (size_t mb : sizes_mb) {
    const uint32_t n = (uint32_t)(mb * 1024 * 1024 / 32);
    std::vector<uint32_t> h((size_t)n * 8);
    uint32_t x = 12345u;
    for (uint32_t i = 0; i < n; ++i) {
      for (int k = 0; k < 8; ++k) {
        x = x * 1664525u + 1013904223u;
        h[(size_t)i * 8 + k] = x;
      }
      // links: a random record; with 64-byte fetches the record index is even-aligned below
      h[(size_t)i * 8] = (uint32_t)(((uint64_t)(x >> 4) * 2654435761ull) % (n - 1));
    }
    uint4 *d_table;
    CK(hipMalloc(&d_table, (size_t)n * 32 + 64));
    CK(hipMemcpy(d_table, h.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    for (int waves_per_simd : {2, 4, 5, 6, 8}) {
      for (uint32_t active : {16u, 32u, 64u}) {
        for (int mode : {16, 32, 64, 2}) {
          // occupancy limit through dynamic LDS: a CU holds 160 KB
          const int blocks_per_cu = waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD
          const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 512;
          const int grid = cus * blocks_per_cu;
          auto launch = [&]() {
            if (mode == 16) hipLaunchKernelGGL(k_chase<16>, dim3(grid), dim3(256), lds, 0, d_table, n, steps, active, d_out);
            if (mode == 32) hipLaunchKernelGGL(k_chase<32>, dim3(grid), dim3(256), lds, 0, d_table, n, steps, active, d_out);
            if (mode == 64) hipLaunchKernelGGL(k_chase<64>, dim3(grid), dim3(256), lds, 0, d_table, n, steps, active, d_out);
            if (mode == 2) hipLaunchKernelGGL(k_chase2, dim3(grid), dim3(256), lds, 0, d_table, n, steps, active, d_out);
          };
          for (auto f : {&k_chase<16>, &k_chase<32>, &k_chase<64>})
            CK(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
          CK(hipFuncSetAttribute((const void *)k_chase2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
          launch();
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          launch();
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          const double lanes = (double)grid * 4.0 * active;
          const double chains = mode == 2 ? 2.0 : 1.0;
          const double lane_steps = lanes * steps * chains;
          const double ns_per_step = ms * 1e6 / steps;  // latency of one dependent step as seen by a lane
          printf("table %4zu MB  waves/SIMD %d  lanes %2u  %s  %8.1f G lane-steps/s  %7.0f ns/step  %6.1f GB/s useful\n", mb,
                 waves_per_simd, active, mode == 2 ? "2x32B" : (mode == 16 ? "  16B" : (mode == 32 ? "  32B" : "  64B")),
                 lane_steps / (ms * 1e6), ns_per_step, lane_steps * (mode == 2 ? 32 : mode) / (ms * 1e6));
          fflush(stdout);
        }
      }
    }
    CK(hipFree(d_table));
  }
  return 0;
}
