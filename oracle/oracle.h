/*
 * oracle.h — CPU oracle of the wavefront path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * kernels (shaders/{random,queue,sample,ray_extend,shade}.slang and the
 * per-material / connect kernels under shaders/archive/) used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  No
 * product code includes, links or calls it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures
 * for this path and cannot be built or run here (it needs cargo, slangc and a
 * Vulkan driver), so this restatement is pinned only by known-answer values
 * derived from the text of the reference's formulas (tests/golden/).
 */
#ifndef MIRT_ORACLE_H
#define MIRT_ORACLE_H

#include "mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

/* Validates the flat buffers and precomputes each instance's matrices.
 * The arrays are copied.  Returns NULL on invalid input. */
oracle_scene *oracle_scene_create(const MirtSceneDesc *desc);
void oracle_scene_free(oracle_scene *s);

/* random.slang: advance `state` once, return the value. */
float oracle_rng_next(uint32_t state[4]);
/* The per-sample stream key -> initial state. */
void oracle_rng_init(uint32_t seed, uint32_t source, uint32_t sample_no, uint32_t state[4]);

/* ray_extend.slang closest hit / any hit over a ray set (OpenMP over rays). */
void oracle_trace_closest(const oracle_scene *s, const MirtRay *rays, size_t n, MirtHitId *out,
                          MirtTraversalCounts *counts);
/* any-hit: out[i] holds the FIRST accepted triangle in traversal order
 * (instance_id 0xFFFFFFFF when nothing lies within [0, tmax[i]]). */
void oracle_trace_any(const oracle_scene *s, const MirtRay *rays, const float *tmax, size_t n,
                      MirtHitId *out, MirtTraversalCounts *counts);
/* Closest hit by testing every triangle of every instance (no BVH). */
void oracle_trace_brute(const oracle_scene *s, const MirtRay *rays, size_t n, MirtHitId *out);

typedef struct OracleRenderStats {
  uint64_t extend_rays, connect_rays, shade_events, finished_samples;
  MirtTraversalCounts extend_counts, connect_counts;
  uint64_t overflowed_sums; /* adds that found a radiance sum at its 32-bit limit (the sum stays there) */
} OracleRenderStats;

/* Render samples [first_sample, first_sample + spp) of every source, one whole
 * path at a time, and ADD them into sum4[4*n_sources] / count[n_sources].
 * threads <= 0 uses the OpenMP default. */
void oracle_render(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                   uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                   uint32_t first_sample, uint32_t spp, uint32_t *sum4, uint32_t *count,
                   OracleRenderStats *stats, int threads);

/* oracle_render over the sources listed in active[0..n_active) only (NULL = all
 * n_active = n_sources of them): spp samples for each, sample numbers as in
 * oracle_render.  moments (may be NULL): per source the sum and the sum of
 * squares of y = min(r + g + b in fixed point, 2^24 - 1), ADDED to. */
void oracle_render_active(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                          uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                          uint32_t first_sample, uint32_t spp, const uint32_t *active, uint32_t n_active,
                          uint32_t *sum4, uint32_t *count, uint64_t *moments, OracleRenderStats *stats, int threads);

/* Adaptive sampling's convergence test; sets bit 0 of flags[i] (SampleFlag::Converged,
 * shaders/common.slang:141-144) and returns the number of sources left unflagged. */
uint32_t oracle_adapt_update(const uint32_t *count, const uint64_t *moments, uint32_t *flags, uint32_t n_sources,
                             float rel_threshold, uint32_t min_samples);

/* The wavefront simulation with the reference's random streams: one per path
 * slot, seeded from slot_rng and never re-seeded (params->rng_mode must be
 * MIRT_RNG_PER_SLOT).  Runs until sample_limit samples have been handed out and
 * the queues are empty, or for n_frames frames.  The slots that end in a frame
 * take the next sample numbers in ascending slot order. */
void oracle_wavefront_per_slot(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                               uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                               const uint32_t (*slot_rng)[4], uint64_t sample_limit, uint64_t n_frames, uint32_t slots,
                               uint32_t *sum4, uint32_t *count, OracleRenderStats *stats);

/* The same samples through a sequential simulation of the wavefront: `slots`
 * path slots, LIFO queues (queue.slang), one bounce per frame in the order
 * sample -> extend -> [connect] -> shade.  Must produce identical sums. */
void oracle_render_wavefront(const oracle_scene *s, const MirtCamera *camera,
                             const MirtSampleSource *sources, uint32_t n_sources, uint32_t width,
                             uint32_t height, const MirtParams *params, uint32_t first_sample,
                             uint32_t spp, uint32_t slots, uint32_t *sum4, uint32_t *count,
                             OracleRenderStats *stats, uint64_t *frames);

/* The first `n_frames` frames of the same simulation with samples that keep
 * regenerating (no sample budget): what n_frames frames of the reference's
 * per-frame dispatch (src/pathtracer_manager.rs:74-124) have accumulated. */
void oracle_step_wavefront(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                           uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                           uint32_t first_sample, uint64_t n_frames, uint32_t slots, uint32_t *sum4,
                           uint32_t *count, OracleRenderStats *stats);

/* The extension rays of bounce `bounce` (0 = camera rays) of sample
 * `sample_no` of every source; out[i].last_inst == 0xFFFFFFFF marks a path
 * that ended earlier. */
void oracle_bounce_rays(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                        uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                        uint32_t sample_no, uint32_t bounce, MirtRay *out);

/* The connect (shadow) ray traced at the hit of bounce `bounce` of sample
 * `sample_no` of every source (per-material shading with light sampling):
 * origin, direction, excluded light triangle in last_inst / last_prim, and the
 * light distance in tmax_out.  out[i].last_inst == 0xFFFFFFFF when no connect
 * ray is traced there. */
void oracle_connect_rays(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                         uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                         uint32_t sample_no, uint32_t bounce, MirtRay *out, float *tmax_out);

/* Tone-map means to packed pixels (sample.slang:18-57, common.slang:177-183). */
void oracle_resolve(const uint32_t *sum4, const uint32_t *count, const MirtSampleSource *sources,
                    uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                    uint32_t *rgb);

/* +,-,*,/,sqrt,fma on the host, same output layout as mirt_math_selftest. */
void oracle_math_selftest(const float *a, const float *b, const float *c, size_t n, float *out6);

#ifdef __cplusplus
}
#endif
#endif
