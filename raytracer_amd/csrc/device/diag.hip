// diag.hip — the measured ceiling the traversal kernels are priced against.
//
// A BVH walk is, per lane, a chain of dependent fetches of small records whose
// addresses come out of the record before: its speed limit is the rate at which
// the memory pipeline (L1 miss path, L2, fabric, Infinity Cache) serves such
// per-lane gathers from a table of the scene's size — not the HBM streaming
// rate.  k_gather_chase measures exactly that, with the block shape of the
// traversal kernels (one 1024-thread block per CU, every lane walking its own
// chain through 32-byte records, two 16-byte loads per step like a node fetch),
// and nothing else going on: no arithmetic, no divergence, no refill.
#include "kernels.h"

namespace mirt {
namespace {

__device__ __forceinline__ uint32_t diag_mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// record i: word 0 = index of the next record, the rest filler.  The links are
// the affine map i -> (i * P + C) mod n with P a prime above 2^31 >= n: a
// permutation, so chains that start on different records never merge (with a
// random map they would, and lanes following each other's chains hit in cache),
// and neighbouring records link to records far apart.
__global__ __launch_bounds__(256) void k_gather_fill(uint4 *table, uint32_t n_records) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_records) return;
  const uint32_t link = (uint32_t)(((unsigned long long)i * 2654435761ull + 1013904223ull) % n_records);
  table[2 * (size_t)i] = make_uint4(link, diag_mix(i), i, 1u);
  table[2 * (size_t)i + 1] = make_uint4(i ^ 0x55555555u, 2u, 3u, diag_mix(~i));
}

__global__ __launch_bounds__(1024) void k_gather_chase(const uint4 *__restrict__ table, uint32_t n_records, uint32_t steps,
                                                       uint32_t *__restrict__ sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (uint32_t)(((unsigned long long)diag_mix(gid + 99u) * n_records) >> 32);
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    const uint4 a = table[2 * (size_t)idx], b = table[2 * (size_t)idx + 1];
    acc += a.y ^ b.w;
    idx = a.x;
  }
  if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive; practically never taken
}

}  // namespace

int gather_ceiling(hipStream_t stream, int num_cus, size_t table_bytes, uint32_t steps, uint32_t repeats,
                   double *lane_steps_per_s, std::string &err) {
  const size_t n = table_bytes / 32u;
  if (n < 1024u || n > 0x7FFFFFFFull || steps == 0u || repeats == 0u) {
    err = "gather ceiling: table of 32 KiB to 64 GiB, steps and repeats > 0";
    return MIRT_ERR_INVALID_ARGUMENT;
  }
  uint4 *table = nullptr;
  uint32_t *sink = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&table), n * 32u);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&sink), 256);
  if (e == hipSuccess) e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  double best = 0.0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_gather_fill, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, stream, table, (uint32_t)n);
    const dim3 grid((unsigned)num_cus);
    hipLaunchKernelGGL(k_gather_chase, grid, dim3(1024), 0, stream, table, (uint32_t)n, steps, sink);  // warm-up
    for (uint32_t r = 0; r < repeats && e == hipSuccess; ++r) {
      (void)hipEventRecord(a, stream);
      hipLaunchKernelGGL(k_gather_chase, grid, dim3(1024), 0, stream, table, (uint32_t)n, steps, sink);
      (void)hipEventRecord(b, stream);
      e = hipStreamSynchronize(stream);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
      if (e == hipSuccess && ms > 0.0f) {
        const double rate = (double)num_cus * 1024.0 * (double)steps / ((double)ms * 1e-3);
        if (rate > best) best = rate;
      }
    }
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (table) (void)hipFree(table);
  if (sink) (void)hipFree(sink);
  if (e != hipSuccess) {
    err = std::string("gather ceiling: ") + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_HIP;
  }
  *lane_steps_per_s = best;
  return MIRT_OK;
}

}  // namespace mirt
