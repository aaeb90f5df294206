// gather_width.hip — how the per-lane dependent-gather rate depends on the record width.
// One 1024-thread block per CU (the traversal kernels' shape), every lane chases its own chain through
// records of W bytes (W / 16 dwordx4 loads per step, all consumed); links are an affine permutation so
// chains never merge.  Prints lane-steps/s and bytes/s for W = 16, 32, 64, 128 and several table sizes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_width tools/gather_width.hip && /tmp/gather_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int Q>  // Q = uint4 per record
__global__ __launch_bounds__(256) void fill(uint4 *table, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t link = (uint32_t)(((unsigned long long)i * 2654435761ull + 1013904223ull) % n);
  for (int q = 0; q < Q; ++q) table[(size_t)Q * i + q] = make_uint4(q == 0 ? link : i * 7u + q, i ^ q, 3u * i, q + 1u);
}

template <int Q>
__global__ __launch_bounds__(1024) void chase(const uint4 *__restrict__ table, uint32_t n, uint32_t steps, uint32_t *sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (uint32_t)(((unsigned long long)(gid * 2246822519u + 99u) * n) >> 32);
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    uint4 r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = table[(size_t)Q * idx + q];
#pragma unroll
    for (int q = 1; q < Q; ++q) acc += r[q].y ^ r[q].w ^ r[q].x;
    acc += r[0].z;
    idx = r[0].x;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// Two neighbouring lanes chase ONE chain through 32-byte records: lane parity h loads half h of the record (the two
// accesses fall into one 64-byte line and one instruction), the link goes from the even lane to the odd one with a
// data-parallel move.  Half as many chains per wave, one load instruction per step instead of two.
__global__ __launch_bounds__(1024) void chase_pairs(const uint4 *__restrict__ table, uint32_t n, uint32_t steps, uint32_t *sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chain = gid >> 1, half = gid & 1u;
  uint32_t idx = (uint32_t)(((unsigned long long)(chain * 2246822519u + 99u) * n) >> 32);
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    const uint4 r = table[(size_t)2 * idx + half];
    acc += r.y ^ r.w ^ r.z;
    // quad_perm [0, 0, 2, 2]: the even lane's word in both lanes of the pair
    idx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.x, 0xA0, 0xF, 0xF, false);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

void run_pairs(size_t table_bytes, int cus) {
  const uint32_t n = (uint32_t)(table_bytes / 32u);
  uint4 *table; uint32_t *sink;
  CHECK(hipMalloc((void **)&table, (size_t)n * 32u));
  CHECK(hipMalloc((void **)&sink, 256));
  hipLaunchKernelGGL(fill<2>, dim3((n + 255u) / 256u), dim3(256), 0, 0, table, n);
  const uint32_t steps = 2000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(chase_pairs, dim3(cus), dim3(1024), 0, 0, table, n, steps, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double rate = (double)cus * 512.0 * steps / (ms * 1e-3);
    if (r > 0 && rate > best) best = rate;
  }
  printf("table %6.0f MiB  record  32 B, two lanes to a chain  %7.1f G record-steps/s  %8.1f GB/s\n", table_bytes / 1048576.0,
         best * 1e-9, best * 32.0 * 1e-9);
  fflush(stdout);
  CHECK(hipFree(table)); CHECK(hipFree(sink));
}

template <int Q>
void run(size_t table_bytes, int cus) {
  const uint32_t n = (uint32_t)(table_bytes / (16u * Q));
  uint4 *table; uint32_t *sink;
  CHECK(hipMalloc((void **)&table, (size_t)n * 16u * Q));
  CHECK(hipMalloc((void **)&sink, 256));
  hipLaunchKernelGGL(fill<Q>, dim3((n + 255u) / 256u), dim3(256), 0, 0, table, n);
  const uint32_t steps = 2000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(chase<Q>, dim3(cus), dim3(1024), 0, 0, table, n, steps, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double rate = (double)cus * 1024.0 * steps / (ms * 1e-3);
    if (r > 0 && rate > best) best = rate;
  }
  printf("table %6.0f MiB  record %3d B  %7.1f G lane-steps/s  %8.1f GB/s\n", table_bytes / 1048576.0, 16 * Q, best * 1e-9,
         best * 16.0 * Q * 1e-9);
  fflush(stdout);
  CHECK(hipFree(table)); CHECK(hipFree(sink));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  for (size_t mib : {2, 16, 64, 192}) {
    run<1>(mib << 20, cus); run<2>(mib << 20, cus); run_pairs(mib << 20, cus); run<4>(mib << 20, cus); run<6>(mib << 20, cus);
    run<8>(mib << 20, cus);
  }
  return 0;
}
