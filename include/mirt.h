/*
 * mirt.h — C ABI of the MI355X wavefront path-tracing core (libmirt).
 *
 * This is the boundary a host application binds instead of the reference's
 * wgpu compute dispatch.  Every entry point names the reference interface it
 * replaces (paths are relative to the reference checkout).  All structs are
 * plain-old-data with the byte layout of the reference's #[repr(C)] bytemuck
 * mirrors, so a Rust caller can pass `Vec<T>::as_ptr()` directly.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MirtStatus;
 *     mirt_last_error() gives the text of the last failure on that context;
 *   - inputs are copied during the call (the caller may free them at once);
 *   - outputs are written into caller-allocated host buffers;
 *   - a context is used by one thread at a time, from any thread;
 *   - nothing here throws or aborts across the boundary.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum MirtStatus {
  MIRT_OK = 0,
  MIRT_ERR_INVALID_ARGUMENT = -1,
  MIRT_ERR_NO_DEVICE = -2,
  MIRT_ERR_OUT_OF_MEMORY = -3,
  MIRT_ERR_HIP = -4,
  MIRT_ERR_NOT_READY = -5, /* scene or state missing */
  MIRT_ERR_BAD_SCENE = -6, /* an index in the scene buffers is out of range */
  MIRT_ERR_INTERNAL = -7   /* an exception inside the library, caught at the boundary (host memory running out
                              reads as MIRT_ERR_OUT_OF_MEMORY) */
} MirtStatus;

/* ------------------------------------------------- scene PODs (bind group 0) */

/* shaders/common.slang:29-33, src/mesh.rs:40-46 (GPUVertexData) */
typedef struct MirtVertex {
  float position[4];
  float normal[4];
  float uv[4];
} MirtVertex; /* 48 B */

/* shaders/scene.slang:15 (uint4 index), src/mesh.rs:28 (UVec4 face) */
typedef struct MirtIndex {
  uint32_t v[4]; /* xyz used, w = 0 */
} MirtIndex; /* 16 B */

/* shaders/bvh.slang:8-16, src/bvh.rs:200-224 (BVHNodeGPU); `right` is the
 * skip link of the threaded traversal (src/bvh.rs:217). */
typedef struct MirtBVHNode {
  float lb[4];
  float ub[4];
  uint32_t left;
  uint32_t right;
  uint32_t is_leaf;
  uint32_t start;
  uint32_t end;
  uint32_t pad[3];
} MirtBVHNode; /* 64 B */

/* shaders/common.slang:171-175, src/mesh.rs:32-38 */
typedef struct MirtGeometryOffsets {
  uint32_t vertex;
  uint32_t index;
  uint32_t blas_node;
} MirtGeometryOffsets; /* 12 B */

/* shaders/common.slang:165-169, src/instance.rs:1-7 */
typedef struct MirtInstance {
  uint32_t transform;
  uint32_t geometry;
  uint32_t material;
} MirtInstance; /* 12 B */

/* shaders/common.slang:87-91, src/transform.rs:4-10; rotation = Euler xyz rad */
typedef struct MirtTransform {
  float scale[4];
  float rotation[4];
  float translation[4];
} MirtTransform; /* 48 B */

/* shaders/common.slang:7-18, src/material.rs:12-25.  A texture id of 0 means
 * "none" (the binder's own test is `emissive_texture > 0`, src/binder.rs:213-229);
 * id t > 0 names textures[t - 1] of the scene.  colour_texture multiplies
 * colour.rgb and emissive_texture multiplies emissive.rgb at the hit's
 * interpolated vertex uv (glTF's factor x texture rule); the other two ids are
 * carried and not read. */
typedef struct MirtMaterial {
  uint32_t colour_texture;
  uint32_t emissive_texture;
  uint32_t metallic_roughness_texture;
  uint32_t normal_texture;
  float colour[4];
  float emissive[4];
  float metallic;
  float roughness;
  float ior;
  float transmission;
} MirtMaterial; /* 64 B */

/* shaders/common.slang:153-160 (std140), src/camera.rs:102-115 (first 64 B) */
typedef struct MirtCamera {
  float position[3];
  uint32_t pad0;
  float forward[3];
  uint32_t pad1;
  float up[3];
  uint32_t pad2;
  float dims[2];
  float focal_length;
  uint32_t changed;
} MirtCamera; /* 64 B */

/* shaders/common.slang:146-151, src/pathtracer_state.rs:64-71 */
typedef struct MirtSampleSource {
  float screen_pos[2];
  uint32_t out_pos[2];
  uint32_t sample_count;
  uint32_t flags;
} MirtSampleSource; /* 24 B */

/* Per-material PODs of the per-material ("classic") shading kernels:
 * src/lambertian.rs, src/metallic.rs, src/dielectric.rs, src/emissive.rs and
 * shaders/archive/{lambertian,metallic,dielectric,emissive}.slang. */
typedef struct MirtLambertianData {
  float albedo[4];
} MirtLambertianData; /* 16 B */
typedef struct MirtMetallicData {
  float albedo[4];
  float fuzz;
  uint32_t pad[3];
} MirtMetallicData; /* 32 B */
typedef struct MirtDielectricData {
  float albedo[4];
  float ir;
  uint32_t pad[3];
} MirtDielectricData; /* 32 B */
typedef struct MirtEmissiveData {
  float albedo[4];
} MirtEmissiveData; /* 16 B */

/* Material kind routing of shaders/archive/logic.slang:170-189 (hit.mat). */
enum {
  MIRT_MAT_NONE = 0,
  MIRT_MAT_LAMBERTIAN = 1,
  MIRT_MAT_METALLIC = 2,
  MIRT_MAT_DIELECTRIC = 3,
  MIRT_MAT_EMISSIVE = 4
};

/* One entry per slot of the `materials` array: which per-material kernel
 * shades it and which row of that kernel's data table it uses
 * (archive Instance.material / Instance.material_data,
 * shaders/archive/tlas.slang:31-36). */
typedef struct MirtClassicMaterialRef {
  uint32_t kind;
  uint32_t data;
} MirtClassicMaterialRef; /* 8 B */

/* A texture: width x height RGBA8 texels (R | G<<8 | B<<16 | A<<24, row by
 * row, row 0 at v = 0) starting at texels[first_texel]; addressing repeats.
 * The reference carries texture ids in Material and never binds a texture
 * (src/material.rs:15-18; its glTF loader is commented out,
 * src/scenes.rs:332-399), so the sampling rule is this library's:
 * channel / 255, nearest or bilinear with texel centres at (i + 0.5) / size. */
enum { MIRT_FILTER_NEAREST = 0, MIRT_FILTER_BILINEAR = 1 };
typedef struct MirtTexture {
  uint32_t width, height;
  uint32_t first_texel;
  uint32_t filter; /* MIRT_FILTER_* */
} MirtTexture; /* 16 B */

/* The ten read-only scene buffers in bind-group-0 order
 * (shaders/scene.slang:10-29; built by src/binder.rs:325-370 and
 * src/mesh.rs:198-284), plus the optional per-material tables. */
typedef struct MirtSceneDesc {
  const MirtInstance *instances;              uint32_t n_instances;       /* binding 0 */
  const MirtGeometryOffsets *geometry_offsets; uint32_t n_geometries;     /* binding 1 */
  const MirtVertex *vertices;                 uint32_t n_vertices;        /* binding 2 */
  const MirtIndex *indices;                   uint32_t n_indices;         /* binding 3 */
  const MirtBVHNode *blas_nodes;              uint32_t n_blas_nodes;      /* binding 4 */
  const MirtMaterial *materials;              uint32_t n_materials;       /* binding 5 */
  const MirtTransform *transforms;            uint32_t n_transforms;      /* binding 6 */
  const MirtBVHNode *tlas_nodes;              uint32_t n_tlas_nodes;      /* binding 7 */
  const uint32_t *tlas_to_instances;          uint32_t n_tlas_to_instances; /* binding 8 */
  const uint32_t *light_sources;              uint32_t n_light_sources;   /* binding 9 (instance ids; [0xFFFFFFFF] = none) */
  /* classic shading only; may all be NULL/0 when shade_mode == MIRT_SHADE_UBER */
  const MirtClassicMaterialRef *classic_refs; uint32_t n_classic_refs;    /* one per material */
  const MirtLambertianData *lambertian;       uint32_t n_lambertian;
  const MirtMetallicData *metallic;           uint32_t n_metallic;
  const MirtDielectricData *dielectric;       uint32_t n_dielectric;
  const MirtEmissiveData *emissive;           uint32_t n_emissive;
  /* textures named by MirtMaterial.colour_texture / emissive_texture; may be NULL/0 */
  const MirtTexture *textures;                uint32_t n_textures;
  const uint32_t *texels;                     uint32_t n_texels;
} MirtSceneDesc;

/* --------------------------------------------------------------- parameters */
enum { MIRT_SHADE_UBER = 0, MIRT_SHADE_CLASSIC = 1 };
/* MIRT_RNG_PER_SAMPLE: every sample's stream is keyed by (seed, source,
 * sample number): results do not depend on path slot, queue order or GPU count.
 * MIRT_RNG_PER_SLOT: the reference's scheme — one persistent stream per path
 * slot, seeded by the host (src/pathtracer_state.rs:98-110, set with
 * mirt_state_set_rng) and advanced by whatever path runs in the slot
 * (shaders/random.slang:32-36).  Which sample a slot takes next then changes
 * the picture; the reference leaves that to the arrival order of its atomics,
 * here the slots that ended in a frame take the next sample numbers in
 * ascending slot order, one of the runs the reference can make and the same
 * on every run (the sample stage is then three launches over the whole grid,
 * within a few per cent of the default's frame time).  One GPU: the streams
 * belong to slots, so sample numbers cannot be split over devices. */
enum { MIRT_RNG_PER_SAMPLE = 0, MIRT_RNG_PER_SLOT = 1 };

/* Every literal of the reference kernels that a configuration changes. */
typedef struct MirtParams {
  uint32_t max_bounces;   /* shaders/sample.slang:71 (128); archive/logic.slang:119 (8) */
  float sky_radiance;     /* shaders/ray_extend.slang:36 (10.0) */
  float fixed_point_scale;/* shaders/sample.slang:45-47 (1000.0); archive 1024.0 */
  float exposure;         /* shaders/sample.slang:54, exp2(-2.5) */
  uint32_t shade_mode;    /* MIRT_SHADE_* */
  uint32_t rng_mode;      /* MIRT_RNG_* */
  uint32_t seed;          /* RNG stream seed (keyed with source and sample number) */
  uint32_t nee;           /* classic mode: 1 = trace connect (shadow) rays */
  uint32_t adaptive;      /* 1 = keep per-source second moments (the reference's unused sample_std,
                             shaders/pathtracer.slang:31) so that mirt_adapt_update can flag sources */
  uint32_t stage_kernels; /* classic mode with nee: 1 = behind the connect kernel, a routing pass and one
                             kernel per material, each popping its own queue, as the archive frame dispatches
                             its material phases (src/app.rs:402-410); 0 = the hit material's shade event in
                             one kernel over the connect queue.  Same results either way. */
} MirtParams;

/* --------------------------------------------------------------- ray-set API */
typedef struct MirtRay {
  float pos[3];
  float dir[3];
  uint32_t last_inst; /* 0x7FFFFFFF = none (shaders/sample.slang:83-84) */
  uint32_t last_prim;
} MirtRay; /* 32 B */

typedef struct MirtHitId {
  uint32_t instance_id; /* 0xFFFFFFFF on a miss */
  uint32_t triangle_id; /* BLAS-local index after the BVH build's reorder */
  float t;              /* world-space parameter (FLT_MAX on a miss) */
  uint32_t front_face;
} MirtHitId; /* 16 B */

/* Traversal work done for a ray set, in the units of the algorithmic-byte
 * model (64 B per node visit, 76 B per instance visit, 160 B per triangle). */
typedef struct MirtTraversalCounts {
  uint64_t rays;
  uint64_t tlas_nodes;
  uint64_t instance_visits;
  uint64_t blas_nodes;
  uint64_t triangles;
} MirtTraversalCounts;

typedef struct MirtStats {
  uint64_t extend_rays;    /* successful pops of the extension queue */
  uint64_t connect_rays;   /* successful pops of the connect queue */
  uint64_t shade_events;
  uint64_t spawned_samples;
  uint64_t finished_samples;
  uint64_t frames;
  /* per-kernel launch counts and summed HIP-event durations (ms) */
  uint64_t extend_launches;   double extend_ms;   /* the closest-hit walk */
  uint64_t finish_launches;   double finish_ms;   /* its epilogue: hit records and queue routing */
  uint64_t connect_launches;  double connect_ms;
  uint64_t shade_launches;    double shade_ms;
  uint64_t sample_launches;   double sample_ms;
  MirtTraversalCounts extend_counts;  /* filled only when counting is enabled */
  MirtTraversalCounts connect_counts;
  /* of extend_counts.triangles / connect_counts.triangles: the tests made in
   * whole-wave scans of big leaves, split by where the scan read the triangles
   * (the LDS leaf cache: no memory traffic; lane-interleaved blocks in memory,
   * 36 B per triangle).  The remaining tests fetched one 48-B record each. */
  uint64_t extend_tri_wave_lds, extend_tri_wave_mem;
  uint64_t connect_tri_wave_lds, connect_tri_wave_mem;
  /* radiance sums (channels) that stand at the 32-bit limit when the statistics
   * are read: a sum ends on 2^32 - 1 instead of wrapping as the reference's
   * InterlockedAdd does (shaders/sample.slang:45-47) — inside, the sums are
   * 64-bit and are clamped wherever they are handed out.  Non-zero means the
   * image is clipped there: render fewer samples per accumulator or lower
   * fixed_point_scale.  Not cleared by mirt_stats_reset: it describes the sums. */
  uint64_t overflowed_sums;
  /* diagnostic builds only (-DMIRT_PHASE_TIMING), summed over the waves of the
   * extend kernel (times in ticks of the 100 MHz constant clock): [0] service
   * points, [1] lane node steps, [2] service-point time with 5..16 lanes walking,
   * [3] whole-wave leaves, [4] wave time, [5] step iterations, [6] walking lanes
   * summed over iterations, [7] time in whole-wave scans, [8] the slowest wave's
   * time (a maximum, not a sum), [9] step iterations with at most 4 lanes
   * walking, [10] step-loop time with at most 4, [11] service-point time with at
   * most 4, [12] step iterations with 5..16, [13] step-loop time with 5..16,
   * [14] service-point time, [15] step-loop time. */
  uint64_t engine_phase[16];
} MirtStats;

typedef struct mirt_ctx mirt_ctx;

/* ---------------------------------------------------------------- lifecycle */

/* Replaces the Vulkan device/queue set-up of src/render_resources.rs:37-116. */
int mirt_create(int device_ordinal, mirt_ctx **out_ctx);
int mirt_destroy(mirt_ctx *ctx);
const char *mirt_last_error(const mirt_ctx *ctx);
const char *mirt_status_string(int status);
void mirt_default_params(MirtParams *out);

/* Replaces the buffer creation + bind group 0 of src/binder.rs:292-370 and
 * src/mesh.rs:198-284.  Validates every index, then lays the scene out in
 * HBM in the traversal layout. */
int mirt_scene_upload(mirt_ctx *ctx, const MirtSceneDesc *scene);

/* Replaces the TLAS regeneration of binder_system when only transforms,
 * instances, materials or lights changed (src/binder.rs:199-206,257-282): the
 * caller passes the whole scene again and promises that the geometry arrays
 * (geometry_offsets, vertices, indices, blas_nodes) are those of the last
 * mirt_scene_upload; they are not read.  Instances, transforms, materials,
 * TLAS nodes, tlas_to_instances, lights and the per-material tables are
 * validated and replaced; nodes and triangles stay where they are in HBM.
 * Falls back to a full upload if the number of geometries changed or the TLAS
 * outgrew the room reserved for it. */
int mirt_scene_update_instances(mirt_ctx *ctx, const MirtSceneDesc *scene);

/* The same regeneration when ONLY transforms moved (binder_system's
 * transform.is_changed() case, src/binder.rs:199-206): instances, materials,
 * lights and geometry are those of the last upload, `transforms` replaces its
 * transforms array (same count).  Nothing but the instance matrices (formed on
 * the host) crosses PCIe: instance boxes, TLAS::new (src/tlas.rs:61-158) and its
 * packing into the traversal layout run on the device, over the old TLAS in
 * place.  device_ms (may be NULL) receives the device time of build + packing.
 * The result is what mirt_scene_update_instances gives for the host-built TLAS
 * of the same transforms.  Fails (nothing changed) when the new TLAS has more
 * nodes than the room reserved at upload: 1.5 x the uploaded TLAS + 16, and
 * (MIRT_ERR_INVALID_ARGUMENT, before any launch) when the uploaded TLAS does not
 * list exactly one slot per instance: the rebuilt tree lists every instance. */
int mirt_scene_update_transforms(mirt_ctx *ctx, const MirtTransform *transforms,
                                 uint32_t n_transforms, float *device_ms);

/* Replaces PathtracerState::new (src/pathtracer_state.rs:96-353) and
 * PathtracerOutput::new (src/pathtracer.rs:56-122): allocates `threads` path
 * slots, the four queues, the accumulators for `n_sources` sample sources and
 * the width*height packed-RGB output.  A slot takes 96 bytes of device memory
 * with uber shading and 180 with per-material shading and light sampling;
 * throughput grows with `threads` up to about thirty-two slots per pixel at
 * 1080p (launches of 8.4 M rays and more take a faster form of the walk). */
int mirt_state_create(mirt_ctx *ctx, uint32_t width, uint32_t height,
                      uint32_t threads, const MirtSampleSource *sources,
                      uint32_t n_sources, const MirtParams *params);

/* MIRT_RNG_PER_SLOT only: the host's random-state vector, one uint4 per path
 * slot (PathtracerState::new, src/pathtracer_state.rs:98-110: every word in
 * [1000, 2^32)).  n_slots must equal `threads`; required before rendering. */
int mirt_state_set_rng(mirt_ctx *ctx, const uint32_t (*rng)[4], uint32_t n_slots);

/* Replaces Camera::update (src/camera.rs:193-199).  `changed != 0` resets the
 * accumulators like sampleCleanup (shaders/sample.slang:123-145). */
int mirt_camera_set(mirt_ctx *ctx, const MirtCamera *camera);

/* Zero sums and counts and restart sample numbering (sampleCleanup). */
int mirt_reset_accum(mirt_ctx *ctx);

/* Restrict rendering to samples [first_sample, first_sample + n) of every
 * source (multi-GPU partition of samples-per-pixel). */
int mirt_set_sample_range(mirt_ctx *ctx, uint32_t first_sample);

/* ---------------------------------------------------------------- rendering */

/* Replaces pathtracer_phase_execute (src/pathtracer_manager.rs:74-124): run
 * `n_frames` wavefront frames (sample -> extend -> [connect] -> shade), each
 * advancing every in-flight path by one bounce; samples keep regenerating.
 * Returns before the work completes. */
int mirt_step(mirt_ctx *ctx, uint32_t n_frames);

/* Render exactly `spp` further samples for every source and drain the
 * wavefront (offline form of the same loop).  Synchronises. */
int mirt_render_spp(mirt_ctx *ctx, uint32_t spp);

int mirt_synchronize(mirt_ctx *ctx);

/* Replaces the buffer->texture copy of src/pathtracer.rs:124-146: tone-map
 * the means and return width*height packed pixels (R | G<<8 | B<<16). */
int mirt_read_output(mirt_ctx *ctx, uint32_t *rgb, size_t n_pixels);

/* Raw accumulators in the reference's format (shaders/pathtracer.slang:30):
 * sum4[4*n] (x,y,z fixed-point, w unused) and count[n], indexed by sample
 * source.  Inside, each sum is a 64-bit word (it cannot wrap); what is handed
 * out here, to mirt_copy_accum_to_device and to the resolve is
 * min(sum, 2^32 - 1). */
int mirt_read_accum(mirt_ctx *ctx, uint32_t *sum4, uint32_t *count,
                    size_t n_sources);

/* ------------------------------------------------------ adaptive sampling
 * The reference declares the hooks and never uses them: SampleFlag::Converged
 * (shaders/common.slang:141-144), SampleSource.flags, the sample_std buffer
 * (shaders/pathtracer.slang:31), and a spawn that skips a flagged source
 * (shaders/archive/logic.slang:113-116).  With MirtParams.adaptive = 1 every
 * finished sample also adds y = r + g + b (fixed point, clamped to 24 bits) and
 * y^2 to two 64-bit integers per source.  mirt_adapt_update sets flag bit 0 of
 * every source with at least min_samples samples whose mean's standard error
 * is at most rel_threshold x the mean, and from then on render calls hand
 * samples to the unflagged sources only: mirt_render_spp(spp) renders spp
 * further samples for each of THOSE.  Sources flagged in the array given to
 * mirt_state_create are skipped from the start.  Flags are cleared by
 * mirt_reset_accum / a changed camera, as sampleCleanup clears them
 * (shaders/sample.slang:137).  Synchronises.  n_active (may be NULL) receives
 * the number of sources still sampled. */
int mirt_adapt_update(mirt_ctx *ctx, float rel_threshold, uint32_t min_samples, uint32_t *n_active);
int mirt_read_flags(mirt_ctx *ctx, uint32_t *flags, size_t n_sources);
/* moments[2*i] = sum of y, moments[2*i+1] = sum of y^2 of source i.  Integers: a
 * multi-GPU host adds them across ranks, writes the totals back on every rank
 * and calls mirt_adapt_update there, which then flags the same sources everywhere. */
int mirt_read_moments(mirt_ctx *ctx, uint64_t *moments, size_t n_sources);
int mirt_write_moments(mirt_ctx *ctx, const uint64_t *moments, size_t n_sources);

/* Copy the accumulators (sum4 as mirt_read_accum hands it out: 16 bytes per
 * source, 16-byte aligned) into caller-owned DEVICE buffers (for an RCCL
 * reduction driven by the caller) and load reduced values back.  A caller that
 * adds sums across ranks widens them, adds, and clamps to 2^32 - 1 before
 * loading, so that the total ends where one GPU's would. */
int mirt_copy_accum_to_device(mirt_ctx *ctx, void *dev_sum4, void *dev_count,
                              size_t n_sources);
int mirt_load_accum_from_device(mirt_ctx *ctx, const void *dev_sum4,
                                const void *dev_count, size_t n_sources);

/* ------------------------------------------------------------ ray-set entry */

/* Closest hit of each ray with the traversal of extensionMain
 * (shaders/ray_extend.slang:19-215).  `counts` may be NULL. */
int mirt_trace_closest(mirt_ctx *ctx, const MirtRay *rays, size_t n,
                       MirtHitId *out, MirtTraversalCounts *counts);

/* mirt_trace_closest for callers that cannot allocate device memory (tight
 * serving loops, harnesses that forbid hipMalloc, code that runs while another
 * stream of the process is being captured into a HIP graph — an allocation there
 * would invalidate the capture; the call itself copies from and to pageable host
 * memory and synchronises its stream, so it cannot be captured itself):
 * same arguments and results, but the device-side staging (rays,
 * results, visit counters, the launch's work cursor) lives in a fixed
 * file-scope device array of the library — 6 MiB + 256 B, held by every GPU the
 * library runs on until the process ends — instead of being allocated per call.
 * At most 131072 rays per call; more is MIRT_ERR_INVALID_ARGUMENT.  Calls no
 * hipMalloc / hipFree.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU would race, so within one process the library serialises
 * them (a per-device lock held for the whole, synchronous, call); two PROCESSES
 * each have their own copy of the array.  No counterpart in the reference. */
int mirt_trace_closest_static(mirt_ctx *ctx, const MirtRay *rays, size_t n,
                              MirtHitId *out, MirtTraversalCounts *counts);

/* Any hit with t in [0, tmax[i]] (shadow / connect query, the role
 * shaders/ray_connect.slang:1-7 describes).  out[i] is the FIRST triangle the
 * threaded traversal accepts (instance_id 0xFFFFFFFF when unoccluded). */
int mirt_trace_any(mirt_ctx *ctx, const MirtRay *rays, const float *tmax,
                   size_t n, MirtHitId *out, MirtTraversalCounts *counts);

/* Time `repeats` launches of the closest-hit kernel on a resident ray set
 * (HIP events on the context's stream); returns the mean ms per launch. */
int mirt_bench_trace_closest(mirt_ctx *ctx, const MirtRay *rays, size_t n,
                             uint32_t repeats, double *ms_per_launch);

/* The ceiling the traversal kernels are measured against: lane-steps per
 * second of per-lane dependent gathers of 32-byte records (the access pattern of
 * a BVH walk and nothing else) from a table of `table_bytes` — 0 = the size of
 * the uploaded scene's node + triangle array — with the traversal kernels' block
 * shape; best of `repeats` launches of `steps` steps per lane.  Allocates the
 * table for the duration of the call.  Diagnostic; no counterpart in the
 * reference, which has no timing code (src/pathtracer_manager.rs:99). */
int mirt_bench_gather(mirt_ctx *ctx, size_t table_bytes, uint32_t steps,
                      uint32_t repeats, double *lane_steps_per_s);

/* Bytes of the uploaded scene's node + triangle array (what a walk gathers
 * from) and of its LDS leaf-cache image. */
int mirt_scene_bytes(mirt_ctx *ctx, uint64_t *node_and_triangle_bytes,
                     uint64_t *leaf_cache_bytes);

/* ------------------------------------------------------------- BLAS build */

/* BLAS::new (src/blas.rs:62-78) with BVH::initialize(4) (src/bvh.rs:52-178) on
 * the device, for the caller that builds a mesh's BLAS at load time
 * (MeshServer, src/mesh.rs:118-145, on a rayon worker in the reference).
 * positions4: n_positions x 4 floats (Mesh::positions); faces4: n_faces x 4
 * indices (Mesh::faces, w = 0), reordered IN PLACE exactly as the reference's
 * partition loop reorders mesh.faces; nodes: room for nodes_capacity records
 * (2 * n_faces always suffices), filled in BVHNodeGPU form with the
 * reference's node numbering and skip links; *n_nodes = nodes written (or
 * needed, when nodes_capacity is too small).  device_ms (may be NULL) gets the
 * time of the build on the device without the copies.  Synchronises. */
int mirt_build_blas(mirt_ctx *ctx, const float *positions4, uint32_t n_positions,
                    uint32_t *faces4, uint32_t n_faces, MirtBVHNode *nodes,
                    uint32_t nodes_capacity, uint32_t *n_nodes, float *device_ms);

/* TLAS::new (src/tlas.rs:61-158) on the device, for binder_system's "regenerate the
 * TLAS when transforms or meshes changed" (src/binder.rs:257-282): the world box of
 * every instance from the eight corners of its geometry's box — geom_aabbs6 holds
 * (lb.xyz, ub.xyz) of each BLAS root, MeshServer::aabbs() — through
 * translate * (Rx * (Ry * Rz)) * scale, then the builder with leaves of one
 * instance.  nodes (capacity 2 * n_instances always suffices) and
 * instance_ids[n_instances] (TLAS::instance_ids, binding 8) come out byte-identical
 * to the host builder's; the matrices are formed on the host with its sin / cos, the
 * boxes and the tree on the device.  Synchronises. */
int mirt_build_tlas(mirt_ctx *ctx, const float *geom_aabbs6, uint32_t n_geometries,
                    const MirtTransform *transforms, uint32_t n_transforms,
                    const MirtInstance *instances, uint32_t n_instances, MirtBVHNode *nodes,
                    uint32_t nodes_capacity, uint32_t *n_nodes, uint32_t *instance_ids,
                    float *device_ms);

/* ------------------------------------------------- several GPUs, one process */

/* A group is one context per device ordinal, driven from one host thread of
 * the caller (the shape the reference's single-process host needs; a
 * process-per-GPU host uses mirt_set_sample_range and the accumulator copies
 * above with its own collective).  The scene, the state and the camera are
 * replicated; a render call splits its samples-per-pixel over the members,
 * which render concurrently; mirt_group_reduce adds the members' integer
 * accumulators into member 0 (device-to-device copies over xGMI, one add
 * kernel per member) and clears the others, after which the image is read
 * from member 0.  Sample numbers are global to the group, so the result is
 * bit-identical to one context rendering the same number of samples.
 * No counterpart in the reference, which drives one wgpu device
 * (src/pathtracer_manager.rs:195-281). */
typedef struct mirt_group mirt_group;

int mirt_group_create(const int *device_ordinals, int n, mirt_group **out_group);
int mirt_group_destroy(mirt_group *grp);
int mirt_group_size(const mirt_group *grp);
/* Member i, for the per-context calls (reads, statistics, ray sets).  NULL when out of range. */
mirt_ctx *mirt_group_member(mirt_group *grp, int i);
/* The message of the last failed group call. */
const char *mirt_group_last_error(const mirt_group *grp);
/* mirt_scene_upload / mirt_scene_update_instances / mirt_state_create /
 * mirt_camera_set / mirt_reset_accum on every member. */
int mirt_group_scene_upload(mirt_group *grp, const MirtSceneDesc *scene);
int mirt_group_scene_update_instances(mirt_group *grp, const MirtSceneDesc *scene);
int mirt_group_state_create(mirt_group *grp, uint32_t width, uint32_t height, uint32_t threads,
                            const MirtSampleSource *sources, uint32_t n_sources,
                            const MirtParams *params);
int mirt_group_camera_set(mirt_group *grp, const MirtCamera *camera);
int mirt_group_reset_accum(mirt_group *grp);
/* Render `spp` further samples for every source: member i takes spp / n of
 * them (the first spp % n members one more), all members at once. */
int mirt_group_render_spp(mirt_group *grp, uint32_t spp);
/* Member 0 += members 1..n-1, whose accumulators are then cleared: the
 * radiance sums (64-bit inside: they cannot wrap), the counts and, with adaptive sampling, the integer
 * moments. */
int mirt_group_reduce(mirt_group *grp);
/* mirt_adapt_update for the group: reduces (as above), flags the converged
 * sources on member 0 from the totals and gives every member the same flags
 * and active list, so the next mirt_group_render_spp skips the same sources on
 * every GPU.  n_active (may be NULL) receives the sources still active. */
int mirt_group_adapt_update(mirt_group *grp, float rel_threshold, uint32_t min_samples,
                            uint32_t *n_active);
/* mirt_scene_update_transforms on every member; device_ms (may be NULL)
 * receives the slowest member's device time. */
int mirt_group_scene_update_transforms(mirt_group *grp, const MirtTransform *transforms,
                                       uint32_t n_transforms, float *device_ms);

/* -------------------------------------------------------------- diagnostics */
int mirt_stats(mirt_ctx *ctx, MirtStats *out);
int mirt_stats_reset(mirt_ctx *ctx);
/* 1 = count node/instance/triangle visits inside the render kernels (slower). */
int mirt_set_counting(mirt_ctx *ctx, int enabled);
/* 1 = bracket every kernel launch with HIP events (needed for *_ms). */
int mirt_set_event_timing(mirt_ctx *ctx, int enabled);

/* Dump the per-slot extension rays that the next extend launch would trace
 * (for building bounce-k ray sets in tests).  Returns the number written. */
int mirt_dump_extension_rays(mirt_ctx *ctx, MirtRay *out, size_t capacity,
                             size_t *n_written);

/* Elementary arithmetic self-test: evaluates +,-,*,/,sqrt,fma on the device
 * for caller-provided operands (bit-parity check of the arithmetic contract). */
int mirt_math_selftest(mirt_ctx *ctx, const float *a, const float *b,
                       const float *c, size_t n, float *out6 /* 6*n */);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
