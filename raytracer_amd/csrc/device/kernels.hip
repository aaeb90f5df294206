// kernels.hip — the wavefront kernels: sample (accumulate + new_ray), extend,
// shade, and the ray-set trace kernels, for gfx950 (wave64).
//
// One frame advances every in-flight path by one bounce:
//   sample  pops the terminate queue: accumulates the finished sample
//           (shaders/sample.slang:35-57) and spawns the next camera ray into
//           the same slot (sample.slang:59-102)                -> extension queue
//   extend  pops the extension queue: closest hit
//           (shaders/ray_extend.slang:19-51)                   -> shade | terminate
//   shade   pops the shade queue: material evaluation and the next bounce
//           (shaders/shade.slang:154-224)                      -> extension | terminate
//
// Queues are index lists.  Popping needs no atomics: entry i of a queue goes to
// global thread i (grid-stride), because no kernel pushes the queue it pops.
// Pushing compacts with wave64 ballots: a wave counts its pushing lanes with
// one ballot, ranks them with mbcnt, and reserves its slice of the tail with a
// single atomic; the streaming kernels first combine their sixteen waves'
// counts in LDS so that a 1024-thread block issues one atomic per queue.
#include <float.h>

#include "kernels.h"
#include "device_util.h"
#include "sample_stage.h"

namespace mirt {

namespace {

// ---------------------------------------------------------------- sample

__global__ __launch_bounds__(kStreamBlock) void k_sample(DState st, DFrameParams fp) {
  __shared__ BlockPush2 sh;
  __shared__ unsigned long long sh_sample_base;
  const uint32_t par = fp.frame & 1u;
  const uint32_t *term = st.qdata[Q_TERM0 + par];
  uint32_t *ext = st.qdata[Q_EXT0 + par];
  const uint32_t count = st.qcount[Q_TERM0 + par];
  const CamBasis cb = cam_basis(fp.cam);

  // The block reserves the sample numbers of all its entries with one atomic:
  // its grid-stride rounds cover full blocks of entries except the last one.
  const uint32_t bs = blockDim.x, stride = gridDim.x * bs;
  if (threadIdx.x == 0u) {
    unsigned long long mine = 0;
    for (uint32_t base = blockIdx.x * bs; base < count; base += stride) mine += count - base < bs ? count - base : bs;
    sh_sample_base = mine ? atomicAdd(st.sample_index, mine) : 0ull;
  }
  __syncthreads();
  unsigned long long next_sample = sh_sample_base;

  unsigned long long finished = 0, spawned = 0;
  for (uint32_t base = blockIdx.x * bs; base < count; base += stride) {
    const uint32_t i = base + threadIdx.x;
    const bool active = i < count;
    uint32_t slot = 0;
    if (active) {
      slot = term[i];
      const float4 a = path_q(st, slot, 0u);
      const uint32_t source = path_w(st, slot, 7u);
      if (source != kNoSample) {
        accumulate_sample(st, fp, source, a);
        finished++;
      }
    }
    // pull the next sample (shaders/sample.slang:64-68) from the block's reservation
    const unsigned long long si = next_sample + threadIdx.x;
    next_sample += bs;
    const bool spawn = active && si < fp.sample_limit;
    if (spawn) {
      spawn_sample(st, fp, cb, slot, si);
      spawned++;
    } else if (active) {
      path_w(st, slot, 7u) = kNoSample;  // the slot retires
    }
    block_push2(sh, &st.qcount[Q_EXT0 + par], ext, spawn, nullptr, nullptr, false, slot);
  }
  // per-wave totals -> stats
  for (int off = 32; off > 0; off >>= 1) {
    finished += __shfl_down(finished, off);
    spawned += __shfl_down(spawned, off);
  }
  if (lane_id() == 0u) {
    if (finished) atomicAdd(&st.stats[S_FINISHED], finished);
    if (spawned) atomicAdd(&st.stats[S_SPAWNED], spawned);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // clear the counters of the queues that are idle during this kernel
    st.qcount[Q_TERM0 + (par ^ 1u)] = 0u;
    st.qcount[Q_EXT0 + (par ^ 1u)] = 0u;
    st.qcount[Q_CONNECT] = 0u;
    st.qcount[Q_LAMBERTIAN] = 0u;
    st.qcount[Q_METALLIC] = 0u;
    st.qcount[Q_DIELECTRIC] = 0u;
    st.qcount[Q_EMISSIVE] = 0u;
    st.qcount[W_EXTEND] = 0u;
    st.qcount[W_CONNECT] = 0u;
    // With nothing on this frame's terminate queue no block of this kernel touches the sample counter or the
    // extension queue, so both can be read here: when the samples still to be handed out cover every entry of the
    // extension queue, no slot can retire in this frame.
    st.qcount[F_ALL_STAY] =
        (fp.respawn != 0u && count == 0u && *st.sample_index + st.qcount[Q_EXT0 + par] <= fp.sample_limit) ? 1u : 0u;
  }
}

// ---- the sample stage with per-slot random streams (MIRT_RNG_PER_SLOT)
//
// A slot's stream is its own, so which sample a slot takes next changes the
// picture.  The reference leaves that to the arrival order of its atomics
// (shaders/sample.slang:66, shaders/queue.slang:18,28): every pairing of the
// slots that ended in a frame with the next sample numbers is a run it can make.
// This picks ONE of them so that runs can be compared: the slots of the
// terminate queue take the next sample numbers in ascending slot order.
// Three launches, every one over the whole grid:
//   k_mark_terminated  accumulates the finished samples and stamps their slots
//   k_ordered_count    counts the stamped slots of every tile of kStreamBlock
//                      consecutive slots; the block that finishes last turns the
//                      counts into exclusive prefixes and settles the frame's
//                      totals (sample counter, extension queue length)
//   k_ordered_spawn    ranks every stamped slot (tile prefix + rank inside the
//                      tile) and spawns sample `base + rank` into it; sample
//                      numbers rise with the slot, so the slots that still get
//                      one are a prefix of the stamped ones and a slot's rank is
//                      also its place behind the bounce rays already on the
//                      extension queue
static_assert(kStreamBlock % 64u == 0u && kStreamBlock >= 64u && kStreamBlock <= 1024u, "whole waves, at most sixteen");
constexpr uint32_t kStreamWaves = kStreamBlock / 64u;

__global__ __launch_bounds__(kStreamBlock) void k_mark_terminated(DState st, DFrameParams fp) {
  const uint32_t par = fp.frame & 1u;
  const uint32_t *term = st.qdata[Q_TERM0 + par];
  const uint32_t count = st.qcount[Q_TERM0 + par];
  unsigned long long finished = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint32_t slot = term[i];
    const uint32_t source = path_w(st, slot, 7u);
    if (source != kNoSample) {
      accumulate_sample(st, fp, source, path_q(st, slot, 0u));
      finished++;
    }
    st.slot_stamp[slot] = fp.stamp;
  }
  for (int off = 32; off > 0; off >>= 1) finished += __shfl_down(finished, off);
  if (lane_id() == 0u && finished) atomicAdd(&st.stats[S_FINISHED], finished);
}

// Words of DState::ordered_ws: the ticket the blocks of k_ordered_count draw, the frame's
// bases for k_ordered_spawn, then one word per tile.
enum OrderedWs { OW_TICKET = 0, OW_EXT_BASE, OW_SAMPLE_BASE_LO, OW_SAMPLE_BASE_HI, OW_TILES = 8 };

__device__ __forceinline__ uint32_t agent_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kStreamBlock) void k_ordered_count(DState st, DFrameParams fp) {
  __shared__ uint32_t wave_sum[kStreamWaves];
  __shared__ uint32_t is_last;
  const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
  const uint32_t n_tiles = (st.slots + kStreamBlock - 1u) / kStreamBlock;
  uint32_t *ws = st.ordered_ws;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // uniform per block
    const uint32_t slot = tile * kStreamBlock + tid;
    const bool ended = slot < st.slots && st.slot_stamp[slot] == fp.stamp;
    const unsigned long long m = __ballot(ended);
    if (lane == 0u) wave_sum[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0u) {
      uint32_t total = 0;
      for (uint32_t w = 0; w < kStreamWaves; ++w) total += wave_sum[w];
      agent_store(&ws[OW_TILES + tile], total);
    }
    __syncthreads();
  }
  // the block that draws the last ticket sees every tile's count
  __threadfence();
  if (tid == 0u) is_last = atomicAdd(&ws[OW_TICKET], 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (is_last == 0u) return;
  __threadfence();
  // exclusive prefix over the tiles: a run of consecutive tiles per thread, the runs' sums scanned over the block
  const uint32_t per = (n_tiles + kStreamBlock - 1u) / kStreamBlock;
  const uint32_t t0 = tid * per < n_tiles ? tid * per : n_tiles;
  const uint32_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  uint32_t mine = 0;
  for (uint32_t t = t0; t < t1; ++t) mine += agent_load(&ws[OW_TILES + t]);
  uint32_t x = mine;  // inclusive scan within the wave
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, d);
    if (lane >= (uint32_t)d) x += y;
  }
  if (lane == 63u) wave_sum[wid] = x;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < kStreamWaves; ++w) {
    const uint32_t c = wave_sum[w];
    if (w < wid) before += c;
    total += c;
  }
  uint32_t run = before + x - mine;
  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t c = agent_load(&ws[OW_TILES + t]);
    agent_store(&ws[OW_TILES + t], run);
    run += c;
  }
  if (tid == 0u) {
    const uint32_t par = fp.frame & 1u;
    // every stamped slot draws a sample number (shaders/sample.slang:64-68), whether or not the call
    // still has a sample for it
    const unsigned long long base = *st.sample_index;
    const unsigned long long room = fp.sample_limit > base ? fp.sample_limit - base : 0ull;
    const uint32_t spawned = (unsigned long long)total < room ? total : (uint32_t)room;
    const uint32_t ext_base = st.qcount[Q_EXT0 + par];  // bounce rays pushed by the last frame's shade kernels
    ws[OW_EXT_BASE] = ext_base;
    ws[OW_SAMPLE_BASE_LO] = (uint32_t)base;
    ws[OW_SAMPLE_BASE_HI] = (uint32_t)(base >> 32);
    *st.sample_index = base + total;
    st.qcount[Q_EXT0 + par] = ext_base + spawned;
    if (spawned) atomicAdd(&st.stats[S_SPAWNED], (unsigned long long)spawned);
    ws[OW_TICKET] = 0u;
    // clear the counters of the queues that are idle during the sample stage
    st.qcount[Q_TERM0 + (par ^ 1u)] = 0u;
    st.qcount[Q_EXT0 + (par ^ 1u)] = 0u;
    st.qcount[Q_CONNECT] = 0u;
    st.qcount[Q_LAMBERTIAN] = 0u;
    st.qcount[Q_METALLIC] = 0u;
    st.qcount[Q_DIELECTRIC] = 0u;
    st.qcount[Q_EMISSIVE] = 0u;
    st.qcount[W_EXTEND] = 0u;
    st.qcount[W_CONNECT] = 0u;
  }
}

__global__ __launch_bounds__(kStreamBlock) void k_ordered_spawn(DState st, DFrameParams fp) {
  __shared__ uint32_t wave_count[kStreamWaves];
  const uint32_t par = fp.frame & 1u;
  uint32_t *ext = st.qdata[Q_EXT0 + par];
  const CamBasis cb = cam_basis(fp.cam);
  const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
  const uint32_t n_tiles = (st.slots + kStreamBlock - 1u) / kStreamBlock;
  const uint32_t *ws = st.ordered_ws;
  const uint32_t ext_base = ws[OW_EXT_BASE];
  const unsigned long long sample_base = (unsigned long long)ws[OW_SAMPLE_BASE_LO] | ((unsigned long long)ws[OW_SAMPLE_BASE_HI] << 32);
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // uniform per block
    const uint32_t slot = tile * kStreamBlock + tid;
    const bool ended = slot < st.slots && st.slot_stamp[slot] == fp.stamp;
    const unsigned long long m = __ballot(ended);
    if (lane == 0u) wave_count[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = ws[OW_TILES + tile];
    for (uint32_t w = 0; w < kStreamWaves; ++w)
      if (w < wid) before += wave_count[w];
    const uint32_t rank = before + rank_in_mask(m);
    const unsigned long long si = sample_base + rank;
    if (ended && si < fp.sample_limit) {
      spawn_sample(st, fp, cb, slot, si);
      ext[ext_base + rank] = slot;  // the spawned slots join the extension queue in slot order
    } else if (ended) {
      path_w(st, slot, 7u) = kNoSample;  // the slot retires
    }
    __syncthreads();
  }
}

// Adaptive sampling: source s counts as converged when it has at least
// `min_samples` samples and the standard error of the mean of its sample value
// y is at most rel_threshold times that mean:
//   (n*S2 - S1^2) / (n^2 (n-1)) <= thr^2 * S1^2 / n^2   <=>   n*S2 - S1^2 <= thr^2 * S1^2 * (n-1)
// with S1, S2 the integer moments; evaluated in IEEE doubles, operation by
// operation (no contraction), so that a CPU gets the same flags.
__global__ __launch_bounds__(256) void k_adapt_update(DState st, double thr2, uint32_t min_samples) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= st.n_sources) return;
  const uint32_t n = st.count[s];
  if (n < min_samples || n < 2u) return;
  const double S1 = (double)st.moments[2 * (size_t)s], S2 = (double)st.moments[2 * (size_t)s + 1];
  const double S1sq = S1 * S1;
  const double lhs = (double)n * S2 - S1sq;
  const double rhs = thr2 * S1sq * (double)(n - 1u);
  if (lhs <= rhs) st.flags[s] |= 1u;
}

// Every slot starts on the terminate queue holding no sample, so the first
// frame only spawns.  (The reference starts with a full terminate queue of
// zeroed samples, which credits `slots` bogus samples to source 0:
// src/queue.rs:14-18, shaders/sample.slang:38.)
__global__ __launch_bounds__(256) void k_init_slots(DState st) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < st.slots) {
    st.qdata[Q_TERM0][i] = i;
    path_w(st, i, 7u) = kNoSample;
  }
  if (i < QW_COUNT) st.qcount[i] = i == Q_TERM0 ? st.slots : 0u;
}

// The start of a render call: slot i takes sample i of the call (as many as the
// call has samples for) and goes straight onto frame 0's extension queue, entry
// i = slot i.  Replaces k_init_slots plus the first pass of the sample kernel
// over a terminate queue that would list every slot.
__global__ __launch_bounds__(256) void k_spawn_first(DState st, DFrameParams fp) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t n = fp.sample_limit < (unsigned long long)st.slots ? (uint32_t)fp.sample_limit : st.slots;
  if (i < n) {
    spawn_sample(st, fp, cam_basis(fp.cam), i, (unsigned long long)i);
    st.qdata[Q_EXT0][i] = i;
  }
  if (i < QW_COUNT) st.qcount[i] = i == Q_EXT0 ? n : 0u;
  if (i == 0u) {
    *st.sample_index = n;
    atomicAdd(&st.stats[S_SPAWNED], (unsigned long long)n);
  }
}

// ---------------------------------------------------------------- extend

// The closest-hit walk of extensionMain (shaders/ray_extend.slang:12-33) for
// every entry of this frame's extension queue.  It leaves the raw result — u, v, the
// triangle id and the instance (kMissInst when nothing was hit) — in the first
// quarter of the slot's ray record (load_pray in device_util.h), next to the
// direction, for the streaming kernel that follows; keeping the epilogue out of this kernel
// keeps its registers and instruction footprint small.

// Diagnostic builds park every wave's own phase sums in the convergence flags (unused without adaptive
// sampling, readable with mirt_read_flags) when those have room for sixteen words per wave of the grid.
__device__ __forceinline__ uint32_t *extend_phase_waves(const DState &st) {
#ifdef MIRT_PHASE_TIMING
  if (st.moments == nullptr && (size_t)st.n_sources >= (size_t)16 * gridDim.x * (kEngineBlock / 64u)) return st.flags;
#else
  (void)st;
#endif
  return nullptr;
}

template <bool COUNT>
__global__ MIRT_ENGINE_BOUNDS void k_extend(DScene sc, DState st, DFrameParams fp) {
  const uint32_t par = fp.frame & 1u;
  const uint32_t *ext = st.qdata[Q_EXT0 + par];
  const uint32_t count = st.qcount[Q_EXT0 + par];
  LaneCounts lc;

  auto fetch = [&](uint32_t slot, RayIn &r) {
    load_pray(st.ray, slot, r.pos, r.dir, r.last_inst, r.last_prim);
    r.tmax = FLT_MAX;
    r.payload = slot;
  };
  auto finish = [&](bool fin, uint32_t slot, bool found, float t, const BestHit &best) {
    if (fin) {
      // over the ray's first quarter (origin and last instance: the walk is done with them)
      float4 *r4 = reinterpret_cast<float4 *>(st.ray + slot);
      (void)t;  // nothing downstream reads t: the hit point comes from the barycentrics
      r4[0] = make_float4(best.u, best.v, __uint_as_float(best.prim), __uint_as_float(found ? best.inst : kMissInst));
    }
  };
  trace_engine<false, COUNT>(sc, WorkSource{&st.qcount[W_EXTEND], count, ext, &st.stats[S_PHASE0], extend_phase_waves(st)}, lc, fetch, finish);
  if (COUNT) flush_counts(st.stats, S_EXT_TLAS, lc);
  if (COUNT) flush_wave_scan_counts(st.stats, S_EXT_TRI_LDS, lc);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&st.stats[S_EXTEND_RAYS], (unsigned long long)count);
}

// ------------------------------------------------------------ shade (uber)

// unitDiskSample (shaders/shade.slang:101-109)
__device__ __forceinline__ void unit_disk_sample(Rng &rng, float &x, float &y) {
  float px, py;
  do {
    px = 2.0f * rng_next(rng) - 1.0f;
    py = 2.0f * rng_next(rng) - 1.0f;
  } while (__builtin_sqrtf(__builtin_fmaf(py, py, px * px)) > 1.0f);
  x = px;
  y = py;
}
// unitSphereSample (shaders/shade.slang:115-124): only its draw count is
// observable, its value feeds dead code in shadeMain.
__device__ __forceinline__ void unit_sphere_sample_draws(Rng &rng) {
  v3 p;
  do {
    const float r1 = rng_next(rng), r2 = rng_next(rng), r3 = rng_next(rng);
    p = V3(r1, r2, r3);
  } while (length3(p) > 1.0f);
}
// cosineHemisphereSample (shaders/shade.slang:130-140)
__device__ __forceinline__ v3 cosine_hemisphere_sample(v3 n, Rng &rng) {
  float dx, dy;
  unit_disk_sample(rng, dx, dy);
  const float z = __builtin_sqrtf(fmax_nn(0.0f, 1.0f - dx * dx - dy * dy));
  const v3 temp = __builtin_fabsf(n.x) > 0.9f ? V3(0.0f, 1.0f, 0.0f) : V3(1.0f, 0.0f, 0.0f);
  const v3 t1 = normalize3(cross3(n, temp));
  const v3 t2 = cross3(n, t1);
  const v3 nn = normalize3(n);
  return V3(__builtin_fmaf(nn.x, z, __builtin_fmaf(t2.x, dy, t1.x * dx)),
            __builtin_fmaf(nn.y, z, __builtin_fmaf(t2.y, dy, t1.y * dx)),
            __builtin_fmaf(nn.z, z, __builtin_fmaf(t2.z, dy, t1.z * dx)));
}
// Only the draws of a cosineHemisphereSample whose value is unused.
__device__ __forceinline__ void cosine_hemisphere_draws(Rng &rng) {
  float dx, dy;
  unit_disk_sample(rng, dx, dy);
}

__device__ __forceinline__ float heaviside(float x) { return x > 0.0f ? 1.0f : 0.0f; }
__device__ __forceinline__ v3 mix3(v3 a, v3 b, float m) {  // shaders/shade.slang:24-26
  const float k = 1.0f - m;
  return V3(k * a.x + m * b.x, k * a.y + m * b.y, k * a.z + m * b.z);
}
// specularBRDF (shaders/shade.slang:40-53)
__device__ __forceinline__ float specular_brdf(v3 wi, v3 wo, v3 n, float alpha) {
  const v3 h = normalize3(add3(wo, wi));
  const float a2 = alpha * alpha;
  const float nh = dot3(n, h);
  float d = a2 * heaviside(nh);
  d /= kPi * pow2f(pow2f(nh) * (a2 - 1.0f) + 1.0f);
  const float nwi = dot3(n, wi), nwo = dot3(n, wo);
  const float nwi2 = pow2f(nwi), nwo2 = pow2f(nwo);
  float v = heaviside(dot3(h, wi)) / (__builtin_fabsf(nwi) + __builtin_sqrtf(a2 + (1.0f - a2) * nwi2));
  v *= heaviside(dot3(h, wo)) / (__builtin_fabsf(nwo) + __builtin_sqrtf(a2 + (1.0f - a2) * nwo2));
  return v * d;
}
// specularBTDF (shaders/shade.slang:59-72); its three lanes are equal
__device__ __forceinline__ float specular_btdf(v3 wi, v3 wo, v3 n, float alpha) {
  const float nwi = dot3(n, wi), nwo = dot3(n, wo);
  const v3 ht = normalize3(add3(sub3(wo, scale3(n, 2.0f * nwi)), wi));
  const float a2 = alpha * alpha;
  const float nht = dot3(n, ht);
  float dt = a2 * heaviside(nht);
  dt /= kPi * pow2f(pow2f(nht) * (a2 - 1.0f) + 1.0f);
  float vt = heaviside(dot3(ht, wi) / nwi);
  vt *= heaviside(dot3(ht, wo) / nwo);
  vt /= __builtin_fabsf(nwi) + __builtin_sqrtf(a2 + (1.0f - a2)) * pow2f(nwi);
  vt /= __builtin_fabsf(nwo) + __builtin_sqrtf(a2 + (1.0f - a2)) * pow2f(nwo);
  return vt * dt;
}
// material() = mix(dielectricBRDF, metallicBRDF, metallic) (shaders/shade.slang:19-99)
__device__ __forceinline__ v3 material_brdf(v3 wi, v3 wo, v3 n, const DMaterial &m) {
  const v3 colour = V3(m.colour[0], m.colour[1], m.colour[2]);
  const float alpha = pow2f(m.roughness);
  const v3 h = normalize3(add3(wo, wi));
  const float fres = pow5f(1.0f - __builtin_fabsf(dot3(wo, h)));
  const float spec = specular_brdf(wi, wo, n, alpha);
  const v3 diffuse = scale3(colour, 1.0f / kPi);
  const float btdf = specular_btdf(wi, wo, n, alpha);
  const v3 base = mix3(diffuse, scale3(colour, btdf), m.transmission);
  const float f0 = pow2f((1.0f - m.ior) / (1.0f + m.ior));
  const float fr = f0 + (1.0f - f0) * fres;
  const v3 dielectric = mix3(base, V3(spec, spec, spec), fr);
  const v3 metallic = V3(spec * (colour.x + (1.0f - colour.x) * fres), spec * (colour.y + (1.0f - colour.y) * fres),
                         spec * (colour.z + (1.0f - colour.z) * fres));
  return mix3(dielectric, metallic, m.metallic);
}

// shadeMain (shaders/shade.slang:143-224) for one slot whose hit record is
// given in registers: emission, the three sampling draws, throughput update,
// the bounce ray.  Returns whether the path goes on.
// The slot's path record comes in and goes out in registers (pa: radiance +
// bounces, pb: throughput + source, pr: stream state); the caller stores it, and
// the bounce ray wi from the hit point, only when the path goes on.
__device__ __forceinline__ bool shade_uber_slot(const DScene &sc, v3 wo, v3 hit_n, uint32_t inst, uint32_t ff,
                                                const BestHit &best, float4 &pa, float4 &pb, float4 &pr, v3 &wi_out) {
  Rng rng{__float_as_uint(pr.x), __float_as_uint(pr.y), __float_as_uint(pr.z), __float_as_uint(pr.w)};
  DMaterial mat = sc.mat[sc.inst[inst].material];
  if ((mat.tex[0] | mat.tex[1]) != 0u) {
    // base-colour and emissive textures multiply the material's factors (src/material.rs:12-25)
    float tu, tv;
    hit_uv(sc, best, tu, tv);
    if (mat.tex[0] != 0u) {
      const v3 c = tex_sample(sc, mat.tex[0], tu, tv);
      mat.colour[0] = mat.colour[0] * c.x; mat.colour[1] = mat.colour[1] * c.y; mat.colour[2] = mat.colour[2] * c.z;
    }
    if (mat.tex[1] != 0u) {
      const v3 c = tex_sample(sc, mat.tex[1], tu, tv);
      mat.emissive[0] = mat.emissive[0] * c.x; mat.emissive[1] = mat.emissive[1] * c.y; mat.emissive[2] = mat.emissive[2] * c.z;
    }
  }

  // s.rad += s.throughput * mat.emissive.rgb
  pa.x = pa.x + pb.x * mat.emissive[0];
  pa.y = pa.y + pb.y * mat.emissive[1];
  pa.z = pa.z + pb.z * mat.emissive[2];
  const float sgn = ff != 0u ? 1.0f : -1.0f;
  const v3 n = V3(hit_n.x * sgn, hit_n.y * sgn, hit_n.z * sgn);

  cosine_hemisphere_draws(rng);   // diffuse_sample: value unused, draws kept
  unit_sphere_sample_draws(rng);  // metallicSample -> unitSphereSample
  const v3 wi = cosine_hemisphere_sample(n, rng);
  const float pdf = __builtin_fabsf(dot3(wi, n)) / kPi;
  const v3 brdf = material_brdf(wi, wo, n, mat);
  const float ndotwi = __builtin_fabsf(dot3(n, wi));
  pb.x = pb.x * (brdf.x * ndotwi * 1.0f / pdf);
  pb.y = pb.y * (brdf.y * ndotwi * 1.0f / pdf);
  pb.z = pb.z * (brdf.z * ndotwi * 1.0f / pdf);
  const uint32_t bounces = __float_as_uint(pa.w) - 1u;
  pa.w = __uint_as_float(bounces);
  pr = make_float4(__uint_as_float(rng.x), __uint_as_float(rng.y), __uint_as_float(rng.z), __uint_as_float(rng.w));
  wi_out = wi;
  return bounces != 0u;
}

// extensionMain's epilogue and shadeMain in one pass over the extension queue
// (uber shading): the hit record stays in registers between the two, a miss
// adds the sky and ends the path.
//
// RESPAWN: a path that ends here is accumulated and its slot takes the next
// sample of the render call at once (accumulateSample + spawnSample of
// shaders/sample.slang:35-102 done in place), so the slot is back on the
// extension queue of the next frame without a trip through the terminate queue
// and the sample kernel.  Streams are keyed per sample and the sums are
// integers, so the accumulators end on the same bits as with the
// three-kernel frame; only which frame a sample is credited in changes, which
// is why mirt_step keeps the three-kernel frame.
#ifndef MIRT_SHADE_WAVES
#define MIRT_SHADE_WAVES 4
#endif
#ifndef MIRT_SHADE_ATTR
#define MIRT_SHADE_ATTR
#endif
template <bool RESPAWN>
__global__ __launch_bounds__(kStreamBlock, MIRT_SHADE_WAVES) MIRT_SHADE_ATTR void k_finish_shade_uber(DScene sc, DState st, DFrameParams fp) {
  __shared__ BlockPush2 sh;
  __shared__ BlockReserve shr;
  const uint32_t par = fp.frame & 1u;
  const uint32_t *ext = st.qdata[Q_EXT0 + par];
  const uint32_t count = st.qcount[Q_EXT0 + par];
  // With enough samples left for every entry, each path that ends here is replaced and every entry stays: the next
  // queue is this one, copied in place — no counting across the block, no atomic, none of their barriers.
  const bool all_stay = RESPAWN && st.qcount[F_ALL_STAY] != 0u;
  uint32_t shaded = 0, finished = 0, spawned = 0;  // of this thread: at most one per round of the loop
  // Every access of this kernel is a miss of its own (slots are scattered over gigabytes of
  // path state), so what it waits for is round trips, not bytes: everything addressed by the
  // slot alone — the traversal's result, the path record, the ray — is requested at once, before
  // any of it is looked at, and the next round's queue entry is requested a round ahead.
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t next_slot = 0;
  if (blockIdx.x * blockDim.x + threadIdx.x < count) next_slot = ext[blockIdx.x * blockDim.x + threadIdx.x];
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {  // uniform per block
    const uint32_t i = base + threadIdx.x;
    const bool active = i < count;
    const uint32_t slot = next_slot;
    if (i + stride < count) next_slot = ext[i + stride];
    bool alive = false;
    float4 rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // of a path that ends here
    uint32_t source = kNoSample;
    if (active) {
      const float4 *r4 = reinterpret_cast<const float4 *>(st.ray + slot);
      float4 pa = path_q(st, slot, 0u), pb = path_q(st, slot, 1u), pr = path_q(st, slot, 2u);
      const float4 raw = r4[0], rb = r4[1];  // the walk's result, the incoming direction
      BestHit best;
      best.u = raw.x;
      best.v = raw.y;
      best.t = 0.0f;
      best.prim = __float_as_uint(raw.z);
      best.inst = __float_as_uint(raw.w);
      source = __float_as_uint(pb.w);
      if (best.inst == kMissInst) {
        // rad += throughput * sky
        const float sky = fp.p.sky_radiance;
        pa.x = pa.x + pb.x * sky;
        pa.y = pa.y + pb.y * sky;
        pa.z = pa.z + pb.z * sky;
        rad = pa;
        if (!RESPAWN) path_q(st, slot, 0u) = pa;
      } else {
        const v3 wo = V3(rb.x, rb.y, rb.z);  // the incoming direction, not negated
        v3 hp, hn, wi;
        uint32_t ff;
        finish_hit(sc, wo, best, hp, hn, ff);
        alive = shade_uber_slot(sc, wo, hn, best.inst, ff, best, pa, pb, pr, wi);
        rad = pa;
        shaded++;
        if (alive || !RESPAWN) {
          // (with RESPAWN a path that ends here is overwritten by its slot's next sample at once)
          path_q(st, slot, 0u) = pa;
          path_q(st, slot, 1u) = pb;
          path_q(st, slot, 2u) = pr;
        }
        // ray.dir = wi; ray.pos = hit position, no epsilon: the primitive just hit is excluded
        // from the next traversal instead
        if (alive) store_pray(st.ray, slot, hp, wi, best.inst, best.prim);
      }
    }
    if (RESPAWN) {
      const bool done = active && !alive;
      if (done) {
        accumulate_sample(st, fp, source, rad);
        finished++;
      }
      const unsigned long long si = block_reserve(shr, st.sample_index, done);
      if (done && si < fp.sample_limit) {
        spawn_sample(st, fp, cam_basis(fp.cam), slot, si);  // the basis is a few products, cheaper than registers held for it
        spawned++;
        alive = true;
      } else if (done) {
        path_w(st, slot, 7u) = kNoSample;  // the slot retires
      }
      if (all_stay) {
        if (active) st.qdata[Q_EXT0 + (par ^ 1u)][i] = slot;
      } else {
        block_push2(sh, &st.qcount[Q_EXT0 + (par ^ 1u)], st.qdata[Q_EXT0 + (par ^ 1u)], active && alive, nullptr, nullptr,
                    false, slot);
      }
    } else {
      block_push2(sh, &st.qcount[Q_EXT0 + (par ^ 1u)], st.qdata[Q_EXT0 + (par ^ 1u)], active && alive,
                  &st.qcount[Q_TERM0 + (par ^ 1u)], st.qdata[Q_TERM0 + (par ^ 1u)], active && !alive, slot);
    }
  }
  if (all_stay && blockIdx.x == 0u && threadIdx.x == 0u) st.qcount[Q_EXT0 + (par ^ 1u)] = count;
  for (int off = 32; off > 0; off >>= 1) {
    shaded += __shfl_down(shaded, off);
    if (RESPAWN) {
      finished += __shfl_down(finished, off);
      spawned += __shfl_down(spawned, off);
    }
  }
  if (lane_id() == 0u) {
    if (shaded) atomicAdd(&st.stats[S_SHADE_EVENTS], (unsigned long long)shaded);
    if (RESPAWN && finished) atomicAdd(&st.stats[S_FINISHED], (unsigned long long)finished);
    if (RESPAWN && spawned) atomicAdd(&st.stats[S_SPAWNED], (unsigned long long)spawned);
  }
}

// ---------------------------------------------------------- accumulator add

// dst += src, four words per thread (both arrays hold a multiple of four words
// or end with a scalar tail).  SATURATE: a sum that would pass 2^32 - 1 stays
// there, as in accumulate_sample, so that the total of several devices is what
// one device would have accumulated.
template <bool SATURATE>
__global__ __launch_bounds__(256) void k_accum_add(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n) {
  auto add = [](uint32_t a, uint32_t b) {
    const uint32_t r = a + b;
    return (SATURATE && r < a) ? 0xFFFFFFFFu : r;
  };
  const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (i + 4u <= n) {
    uint4 a = *reinterpret_cast<const uint4 *>(dst + i);
    const uint4 b = *reinterpret_cast<const uint4 *>(src + i);
    a.x = add(a.x, b.x);
    a.y = add(a.y, b.y);
    a.z = add(a.z, b.z);
    a.w = add(a.w, b.w);
    *reinterpret_cast<uint4 *>(dst + i) = a;
  } else {
    for (size_t k = i; k < n; ++k) dst[k] = add(dst[k], src[k]);
  }
}

// dst += src over 64-bit words (the integer moments of adaptive sampling)
__global__ __launch_bounds__(256) void k_accum_add64(unsigned long long *__restrict__ dst,
                                                     const unsigned long long *__restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// The sums in the reference's format (shaders/pathtracer.slang:30: three fixed-point words and an unused
// one per source), clamped to 2^32 - 1, and back.
__global__ __launch_bounds__(256) void k_export_sums(DState st, uint32_t *__restrict__ dst4) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= st.n_sources) return;
  uint4 o;
  const unsigned long long a = st.sum3[kSumStride * (size_t)i], b = st.sum3[kSumStride * (size_t)i + 1], c = st.sum3[kSumStride * (size_t)i + 2];
  o.x = a > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a;
  o.y = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
  o.z = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
  o.w = 0u;
  reinterpret_cast<uint4 *>(dst4)[i] = o;
}
__global__ __launch_bounds__(256) void k_import_sums(DState st, const uint32_t *__restrict__ src4) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= st.n_sources) return;
  const uint4 v = reinterpret_cast<const uint4 *>(src4)[i];
  st.sum3[kSumStride * (size_t)i] = v.x;
  st.sum3[kSumStride * (size_t)i + 1] = v.y;
  st.sum3[kSumStride * (size_t)i + 2] = v.z;
}
__global__ __launch_bounds__(256) void k_count_saturated(DState st) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  const bool over = i < kSumStride * (size_t)st.n_sources && st.sum3[i] >= 0xFFFFFFFFull;  // reads as 2^32 - 1: at the limit
  const unsigned long long m = __ballot(over);
  if (lane_id() == 0u && m != 0ull) atomicAdd(&st.stats[S_OVERFLOW], (unsigned long long)__popcll(m));
}

// --------------------------------------------------------------- resolve

__device__ __forceinline__ float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

// acesToneMap + packRgb (shaders/sample.slang:18-33,52-56; common.slang:177-183)
// over the final means.  The reference tone-maps inside accumulateSample with
// the count from BEFORE its own increment; the final count is used here.
__global__ __launch_bounds__(256) void k_resolve(DState st, DFrameParams fp) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= st.n_sources) return;
  const MirtSampleSource src = st.sources[i];
  const uint32_t ox = src.out_pos[0], oy = src.out_pos[1];
  if (ox >= st.width || oy >= st.height) return;
  const uint32_t cnt = st.count[i];
  const float denom = fp.p.fixed_point_scale * (float)cnt;
  float rad[3], v[3], c[3], o[3];
  for (int k = 0; k < 3; ++k) {
    const unsigned long long wide = st.sum3[kSumStride * (size_t)i + k];
    const uint32_t sum = wide > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)wide;  // the 32-bit word the reference keeps
    rad[k] = ((float)sum / denom) * fp.p.exposure;
  }
  const float m1[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83770f}};
  const float m2[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
  for (int k = 0; k < 3; ++k) v[k] = m1[k][0] * rad[0] + m1[k][1] * rad[1] + m1[k][2] * rad[2];
  for (int k = 0; k < 3; ++k) {
    const float a = v[k] * (v[k] + 0.0245786f) - 0.000090537f;
    const float b = v[k] * (0.983729f * v[k] + 0.4329510f) + 0.238081f;
    c[k] = a / b;
  }
  for (int k = 0; k < 3; ++k) o[k] = saturate(m2[k][0] * c[0] + m2[k][1] * c[1] + m2[k][2] * c[2]);
  uint32_t r = (uint32_t)(saturate(o[0]) * 255.0f), g = (uint32_t)(saturate(o[1]) * 255.0f),
           b = (uint32_t)(saturate(o[2]) * 255.0f);
  if (cnt == 0u) r = g = b = 0u;
  st.output[(size_t)ox + (size_t)oy * st.width] = (b << 16) | (g << 8) | r;  // .bgr swizzle: red in the low byte
}

// ----------------------------------------------------------------- trace

template <bool ANY, bool COUNT>
__global__ MIRT_ENGINE_BOUNDS void k_trace(DScene sc, const DRay *rays, const float *tmax, uint32_t n,
                                               MirtHitId *out, unsigned long long *counts5, uint32_t *work) {
  LaneCounts lc;
  auto fetch = [&](uint32_t idx, RayIn &r) {
    load_ray(rays, idx, r.pos, r.dir, r.last_inst, r.last_prim);
    r.tmax = ANY ? tmax[idx] : FLT_MAX;
    r.payload = idx;
  };
  auto finish = [&](bool fin, uint32_t idx, bool found, float t, const BestHit &best) {
    if (!fin) return;
    MirtHitId o{0xFFFFFFFFu, 0xFFFFFFFFu, FLT_MAX, 0u};
    if (found) {
      o.instance_id = best.inst;
      o.triangle_id = best.prim;
      o.t = t;
      if (!ANY) {
        v3 pos, dir, hp, hn;
        uint32_t li, lp, ff;
        load_ray(rays, idx, pos, dir, li, lp);
        finish_hit(sc, dir, best, hp, hn, ff);
        o.front_face = ff;
      }
    }
    reinterpret_cast<float4 *>(out)[idx] =
        make_float4(__uint_as_float(o.instance_id), __uint_as_float(o.triangle_id), o.t, __uint_as_float(o.front_face));
  };
  trace_engine<ANY, COUNT>(sc, WorkSource{work, n, nullptr, nullptr, nullptr}, lc, fetch, finish);
  if (COUNT) flush_counts(counts5, 1, lc);
  if (COUNT && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counts5[0], (unsigned long long)n);
}

__global__ void k_math_selftest(const float *a, const float *b, const float *c, uint32_t n, float *out6) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out6[6 * i + 0] = a[i] + b[i];
  out6[6 * i + 1] = a[i] - b[i];
  out6[6 * i + 2] = a[i] * b[i];
  out6[6 * i + 3] = a[i] / b[i];
  out6[6 * i + 4] = __builtin_sqrtf(__builtin_fabsf(a[i]));
  out6[6 * i + 5] = __builtin_fmaf(a[i], b[i], c[i]);
}

}  // namespace

// --------------------------------------------------------------- launchers

void launch_init_slots(const LaunchCfg &cfg, const DState &st) {
  const uint32_t n = st.slots > (uint32_t)QW_COUNT ? st.slots : (uint32_t)QW_COUNT;
  hipLaunchKernelGGL(k_init_slots, dim3((n + 255u) / 256u), dim3(256), 0, cfg.stream, st);
}

void launch_spawn_first(const LaunchCfg &cfg, const DState &st, const DFrameParams &fp) {
  const uint32_t n = st.slots > (uint32_t)QW_COUNT ? st.slots : (uint32_t)QW_COUNT;
  hipLaunchKernelGGL(k_spawn_first, dim3((n + 255u) / 256u), dim3(256), 0, cfg.stream, st, fp);
}

void launch_sample(const LaunchCfg &cfg, const DScene &, const DState &st, const DFrameParams &fp) {
  hipLaunchKernelGGL(k_sample, stream_grid(cfg, k_sample), dim3(kStreamBlock), 0, cfg.stream, st, fp);
}

void launch_sample_ordered(const LaunchCfg &cfg, const DScene &, const DState &st, const DFrameParams &fp) {
  hipLaunchKernelGGL(k_mark_terminated, stream_grid(cfg, k_mark_terminated), dim3(kStreamBlock), 0, cfg.stream, st, fp);
  hipLaunchKernelGGL(k_ordered_count, stream_grid(cfg, k_ordered_count), dim3(kStreamBlock), 0, cfg.stream, st, fp);
  hipLaunchKernelGGL(k_ordered_spawn, stream_grid(cfg, k_ordered_spawn), dim3(kStreamBlock), 0, cfg.stream, st, fp);
}

void launch_adapt_update(const LaunchCfg &cfg, const DState &st, double threshold_squared, uint32_t min_samples) {
  hipLaunchKernelGGL(k_adapt_update, dim3((st.n_sources + 255u) / 256u), dim3(256), 0, cfg.stream, st, threshold_squared,
                     min_samples);
}

// resident waves per SIMD the engine kernels are launched with (blocks per CU x waves per block / 4)
#ifndef MIRT_ENGINE_WAVES
#define MIRT_ENGINE_WAVES 5
#endif
constexpr int kEngineBlocksPerCU = (MIRT_ENGINE_WAVES * 256) / (int)kEngineBlock > 0 ? (MIRT_ENGINE_WAVES * 256) / (int)kEngineBlock : 1;

int trace_grid_blocks(int num_cus) {
  // Resident blocks per CU of the engine kernel with the most registers, asked
  // once per device (the members of a group call this from one host thread each).
  static std::mutex mu;
  static std::unordered_map<int, int> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = per_device.find(dev);
  if (it == per_device.end()) {
    int n = 0, m = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_extend<false>, (int)kEngineBlock, 0) != hipSuccess || n < 1) n = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k_trace<true, true>, (int)kEngineBlock, 0) == hipSuccess && m >= 1 && m < n) n = m;
    // Four to five waves per SIMD are enough to keep the memory path busy; more
    // resident waves only mean fewer rays per wave and a longer drain, so the
    // grid stops at kEngineBlocksPerCU blocks per CU.
    it = per_device.emplace(dev, n > kEngineBlocksPerCU ? kEngineBlocksPerCU : n).first;
  }
  return it->second * num_cus;
}

void launch_extend(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  const dim3 grid(trace_grid_blocks(cfg.num_cus));
  if (fp.counting) hipLaunchKernelGGL((k_extend<true>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_extend<false>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
}

void launch_finish_shade_uber(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  if (fp.respawn) hipLaunchKernelGGL((k_finish_shade_uber<true>), stream_grid(cfg, k_finish_shade_uber<true>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_finish_shade_uber<false>), stream_grid(cfg, k_finish_shade_uber<false>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
}

void launch_accum_add(const LaunchCfg &cfg, uint32_t *dst, const uint32_t *src, size_t n, bool saturate) {
  if (n == 0) return;
  const size_t threads = (n + 3u) / 4u;
  const dim3 grid((unsigned)((threads + 255u) / 256u));
  if (saturate) hipLaunchKernelGGL(k_accum_add<true>, grid, dim3(256), 0, cfg.stream, dst, src, n);
  else hipLaunchKernelGGL(k_accum_add<false>, grid, dim3(256), 0, cfg.stream, dst, src, n);
}

void launch_accum_add64(const LaunchCfg &cfg, unsigned long long *dst, const unsigned long long *src, size_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_accum_add64, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, cfg.stream, dst, src, n);
}

void launch_export_sums(const LaunchCfg &cfg, const DState &st, uint32_t *dst4) {
  hipLaunchKernelGGL(k_export_sums, dim3((st.n_sources + 255u) / 256u), dim3(256), 0, cfg.stream, st, dst4);
}
void launch_import_sums(const LaunchCfg &cfg, const DState &st, const uint32_t *src4) {
  hipLaunchKernelGGL(k_import_sums, dim3((st.n_sources + 255u) / 256u), dim3(256), 0, cfg.stream, st, src4);
}
void launch_count_saturated(const LaunchCfg &cfg, const DState &st) {
  const size_t n = kSumStride * (size_t)st.n_sources;
  hipLaunchKernelGGL(k_count_saturated, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, cfg.stream, st);
}

void launch_resolve(const LaunchCfg &cfg, const DState &st, const DFrameParams &fp) {
  hipLaunchKernelGGL(k_resolve, dim3((st.n_sources + 255u) / 256u), dim3(256), 0, cfg.stream, st, fp);
}

void launch_trace(const LaunchCfg &cfg, const DScene &sc, const DRay *rays, const float *tmax, size_t n,
                  MirtHitId *out, unsigned long long *counts5, uint32_t *work, bool any_hit, bool counting) {
  if (n == 0) return;
  // `work` must be zero: the chunk cursor of this launch
  const size_t want = (n + kChunk - 1) / kChunk / 4 + 1;  // a block's four waves take one chunk each at least
  const int full = trace_grid_blocks(cfg.num_cus);
  const dim3 grid((unsigned)(want < (size_t)full ? want : (size_t)full));
  const uint32_t n32 = (uint32_t)n;
  if (any_hit) {
    if (counting) hipLaunchKernelGGL((k_trace<true, true>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, rays, tmax, n32, out, counts5, work);
    else hipLaunchKernelGGL((k_trace<true, false>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, rays, tmax, n32, out, counts5, work);
  } else {
    if (counting) hipLaunchKernelGGL((k_trace<false, true>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, rays, tmax, n32, out, counts5, work);
    else hipLaunchKernelGGL((k_trace<false, false>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, rays, tmax, n32, out, counts5, work);
  }
}

void launch_math_selftest(hipStream_t stream, const float *a, const float *b, const float *c, size_t n, float *out6) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_math_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, c, (uint32_t)n, out6);
}

}  // namespace mirt
