// kernels.h — launch interface between the C ABI (api.hip) and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <unordered_map>

#include "layout.h"
#include "mirt.h"

namespace mirt {

// Queue ids.  The terminate and extension queues are double-buffered by frame
// parity: a queue is popped in one kernel and its counter is cleared by a
// later kernel that neither pops nor pushes it, so no kernel ever pushes and
// pops the same queue (the reference keeps that invariant too).
enum QueueId {
  Q_TERM0 = 0, Q_TERM1, Q_EXT0, Q_EXT1, Q_CONNECT, Q_LAMBERTIAN, Q_METALLIC, Q_DIELECTRIC, Q_EMISSIVE,
  Q_COUNT,
  // work cursors of the two ray-tracing kernels (chunks handed to waves), kept
  // in the same counter array and cleared by the sample kernel every frame
  W_EXTEND = Q_COUNT, W_CONNECT,
  // set by the sample kernel of a frame: nonzero when the samples still to be handed out cover every entry
  // of the frame's extension queue, i.e. no slot can retire in this frame (see k_finish_shade_uber)
  F_ALL_STAY,
  QW_COUNT
};

// Device counters kept beside the queue counters.
enum StatId {
  S_EXTEND_RAYS = 0, S_CONNECT_RAYS, S_SHADE_EVENTS, S_SPAWNED, S_FINISHED,
  S_EXT_TLAS, S_EXT_INST, S_EXT_BLAS, S_EXT_TRI, S_CON_TLAS, S_CON_INST, S_CON_BLAS, S_CON_TRI,
  S_EXT_TRI_LDS, S_EXT_TRI_MEM, S_CON_TRI_LDS, S_CON_TRI_MEM,  // whole-wave scans, by where they read their triangles
  S_OVERFLOW,  // channel sums at or beyond 2^32 - 1 (handed out as 2^32 - 1), counted when the statistics are read
  S_PHASE0,  // sixteen words written only by diagnostic builds (-DMIRT_PHASE_TIMING)
  S_COUNT = S_PHASE0 + 16
};

// words per source in DState::sum3: r, g, b.  MIRT_SUM_STRIDE=4 pads them so that one source's words share a
// 32-byte sector; measured slower (the streaming kernels +2 to +3 % in time): the footprint is what costs.
#ifndef MIRT_SUM_STRIDE
#define MIRT_SUM_STRIDE 3
#endif
constexpr uint32_t kSumStride = MIRT_SUM_STRIDE;

struct DState {
  DRay *ray;            // [slots] (position + last instance | direction + last triangle; the walk's result over the first half)
  DHit *hit;            // [slots] (connect stage only)
  DPath *path;          // [slots]
  DShadow *shadow;      // [slots] (classic shading)
  uint32_t *qdata[Q_COUNT];  // [slots] each
  uint32_t *qcount;     // [QW_COUNT]
  unsigned long long *stats;         // [S_COUNT]
  unsigned long long *sample_index;  // next sample to hand out in this render call
  const MirtSampleSource *sources;   // [n_sources]
  // radiance sums, three 64-bit words per source (r, g, b in fixed point).  The reference's 32-bit words
  // (shaders/pathtracer.slang:30) are what every boundary hands out — clamped to 2^32 - 1 there
  // (launch_export_sums) — but inside, a sum that cannot wrap is added to with atomics that return nothing.
  unsigned long long *sum3;  // [kSumStride * n_sources]
  uint32_t *count;      // [n_sources]
  uint32_t *output;     // [width * height]
  uint32_t slots, n_sources, width, height;
  // adaptive sampling (allocated when MirtParams.adaptive != 0, else null)
  unsigned long long *moments;  // [2 * n_sources]: sum and sum of squares of the clamped fixed-point sample value
  uint32_t *flags;              // [n_sources]: bit 0 = converged (SampleFlag::Converged)
  const uint32_t *active;       // sources that still take samples, ascending; null = all of them
  uint32_t n_active;            // entries of `active`; n_sources when it is null
  // per-slot random streams (MIRT_RNG_PER_SLOT): the frame a slot was last seen on the terminate queue
  uint32_t *slot_stamp;         // [slots], null in per-sample mode
  uint32_t *ordered_ws;         // [8 + tiles of kStreamBlock slots]: ticket, frame bases, per-tile counts / prefixes
};

struct DFrameParams {
  MirtParams p;
  MirtCamera cam;
  unsigned long long sample_limit;  // samples to hand out in this render call
  uint32_t first_sample;            // sample number of sample_index 0
  uint32_t frame;                   // frame parity selects the queue buffers
  uint32_t have_lights;
  uint32_t counting;
  uint32_t respawn;  // paths that end in the extend epilogue are accumulated and respawned there
  uint32_t stamp;    // per-slot mode: this frame's value in DState::slot_stamp (never 0)
};

// Threads per block of the streaming frame kernels (see block_push2).
#ifndef MIRT_STREAM_BLOCK
#define MIRT_STREAM_BLOCK 1024
#endif
constexpr unsigned kStreamBlock = MIRT_STREAM_BLOCK;

struct LaunchCfg {
  hipStream_t stream;
  int grid_blocks;  // blocks of kStreamBlock threads for the streaming frame kernels
  int num_cus;      // the ray-tracing kernels launch a persistent grid sized from this
};

// Grid of a streaming kernel: the blocks that can be resident at once — one or
// two 1024-thread blocks per CU, by the kernel's registers — or fewer when the
// queues are shorter.  A second round of blocks behind the resident ones only
// adds a tail (the uber shade kernel, one block per CU: 2.31 -> 2.24 ms per launch).
template <class K>
inline dim3 stream_grid(const LaunchCfg &cfg, K kernel) {
  static std::mutex mu;
  static std::unordered_map<const void *, int> cache;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(reinterpret_cast<const void *>(kernel));
    if (it != cache.end()) per_cu = it->second;
  }
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)kStreamBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    std::lock_guard<std::mutex> lock(mu);
    cache[reinterpret_cast<const void *>(kernel)] = per_cu;
  }
  const int cap = per_cu * cfg.num_cus;
  return dim3((unsigned)(cfg.grid_blocks < cap ? cfg.grid_blocks : cap));
}

// Blocks of kEngineBlock threads that fill the device for the traversal engine
// (resident blocks per CU from the occupancy query — one — times CUs).
int trace_grid_blocks(int num_cus);

void launch_sample(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// per-slot random streams: the same stage with the slots of the terminate queue taking sample numbers in ascending slot order
void launch_sample_ordered(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// flags[s] |= 1 where the standard error of source s's mean has fallen to rel_threshold of the mean
void launch_adapt_update(const LaunchCfg &cfg, const DState &st, double threshold_squared, uint32_t min_samples);
void launch_extend(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// per-material shading with light sampling: the extend epilogue (hit records, bounce count, the light sample
// and its shadow ray, routing to the connect queue)
void launch_extend_finish(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// uber shading: the extend epilogue and shadeMain in one kernel
void launch_finish_shade_uber(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// the any-hit walk of the shadow rays on the connect queue
void launch_connect(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// behind it: the light sample's radiance and the hit material's shade event in one kernel ...
void launch_connect_shade(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// ... or (MirtParams.stage_kernels) a routing pass and one kernel per material, each popping its own queue
void launch_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
// per-material shading without light sampling: the extend epilogue and the hit material's shade event in one kernel
void launch_finish_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp);
void launch_resolve(const LaunchCfg &cfg, const DState &st, const DFrameParams &fp);
// dst[i] += src[i] for n words (the integer accumulators of another context)
void launch_accum_add(const LaunchCfg &cfg, uint32_t *dst, const uint32_t *src, size_t n, bool saturate);
void launch_accum_add64(const LaunchCfg &cfg, unsigned long long *dst, const unsigned long long *src, size_t n);
// the sums in the reference's format: dst4[4 * i + c] = min(sum3[3 * i + c], 2^32 - 1), dst4[4 * i + 3] = 0
void launch_export_sums(const LaunchCfg &cfg, const DState &st, uint32_t *dst4);
// ... and back: sum3[3 * i + c] = src4[4 * i + c]
void launch_import_sums(const LaunchCfg &cfg, const DState &st, const uint32_t *src4);
// stats[S_OVERFLOW] += the number of channel sums at or beyond 2^32 - 1
void launch_count_saturated(const LaunchCfg &cfg, const DState &st);
void launch_init_slots(const LaunchCfg &cfg, const DState &st);
// frame 0 of a render call: the first min(slots, sample_limit) samples spawned into slots 0.. (no terminate queue pass)
void launch_spawn_first(const LaunchCfg &cfg, const DState &st, const DFrameParams &fp);

// `work` is a zeroed device word (the launch's chunk cursor).
void launch_trace(const LaunchCfg &cfg, const DScene &sc, const DRay *rays, const float *tmax, size_t n,
                  MirtHitId *out, unsigned long long *counts5, uint32_t *work, bool any_hit, bool counting);
void launch_math_selftest(hipStream_t stream, const float *a, const float *b, const float *c, size_t n, float *out6);

// BLAS::new on the device (build.hip): host arrays in, `faces4` reordered in
// place, nodes in BVHNodeGPU form out.  Face indices must have been checked
// against n_positions.  `workspace` / `workspace_bytes` is device memory the
// caller keeps between builds (grown here with hipMalloc when too small, freed
// by the caller).  Returns a MirtStatus and a message in `err`.
int build_blas_device(hipStream_t stream, const float *positions4, uint32_t n_positions, uint32_t *faces4,
                      uint32_t n_faces, MirtBVHNode *nodes_out, uint32_t nodes_capacity, uint32_t *n_nodes,
                      float *device_ms, void **workspace, size_t *workspace_bytes, std::string &err);

// TLAS::new on the device (build.hip): geom_aabbs6 = (lb3, ub3) per geometry, mats16 = the
// column-major object-to-world matrix of every instance (built on the host with its sin / cos),
// inst_geometry = geometry index per instance.  Fills nodes in BVHNodeGPU form and
// instance_ids[n_instances] (the instance at every leaf slot).  Workspace as for the BLAS.
// With `resident` the result stays on the device: it is packed into the scene's own TLAS block
// (traversal layout) and nodes_out / instance_ids are not written.
struct TlasResident {
  DNode *nodes;        // the scene's TLAS block
  uint32_t capacity;   // nodes it has room for
  uint32_t *big;       // out-of-line leaf sizes, indexed by the leaf's first slot
  uint32_t *to_inst;   // instance of every slot
};
int build_tlas_device(hipStream_t stream, const float *geom_aabbs6, uint32_t n_geometries, const float *mats16,
                      const uint32_t *inst_geometry, uint32_t n_instances, MirtBVHNode *nodes_out, uint32_t nodes_capacity,
                      uint32_t *n_nodes, uint32_t *instance_ids, float *device_ms, void **workspace, size_t *workspace_bytes,
                      std::string &err, const TlasResident *resident = nullptr);

// Lane-steps per second of per-lane dependent 32-byte gathers from a table of
// `table_bytes` (diag.hip): the best of `repeats` launches of `steps` steps per
// lane, one 1024-thread block per CU.  Allocates and frees the table.
int gather_ceiling(hipStream_t stream, int num_cus, size_t table_bytes, uint32_t steps, uint32_t repeats,
                   double *lane_steps_per_s, std::string &err);

}  // namespace mirt
