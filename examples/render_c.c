/* render_c.c — the two C ABIs used from plain C, the way the reference's host
 * would use them through FFI: build one of the benchmark scenes with the host
 * library, upload it, render, read the image back and write a PPM.
 *
 *   cc examples/render_c.c -Iinclude -Lraytracer_amd/lib -lmirt -lmirt_host \
 *      -Wl,-rpath,$PWD/raytracer_amd/lib -o build/render_c
 *   build/render_c <asset_dir> <config 1..5> <spp> <out.ppm> [width height [members]]
 *
 * With `members` > 1 the same render goes through a group of that many contexts
 * (MIRT_GROUP_DEVICES="0,1,..." names their devices; default: all on device 0),
 * which split the samples and add their accumulators up; the output is the same.
 * With MIRT_DEVICE_BUILD=1 in the environment the meshes' BLASes are built on
 * device 0 (mirt_build_blas behind the host model's builder callback) instead of
 * on the host; the output is the same again.
 *
 * Prints "checksum <sum of all fixed-point radiance sums> samples <n>". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mirt.h"
#include "mirt_host.h"

#define CHECK(ctx, call)                                                        \
  do {                                                                          \
    int rc_ = (call);                                                           \
    if (rc_ != 0) {                                                             \
      fprintf(stderr, "%s: %s (%s)\n", #call, mirt_status_string(rc_), (ctx) ? mirt_last_error(ctx) : ""); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

/* mirt_build_blas as the host model's BLAS builder */
static int build_on_device(void *ctx, const float *positions4, uint32_t n_positions, uint32_t *faces4,
                           uint32_t n_faces, MirtBVHNode *nodes, uint32_t capacity, uint32_t *n_nodes) {
  return mirt_build_blas((mirt_ctx *)ctx, positions4, n_positions, faces4, n_faces, nodes, capacity, n_nodes, NULL);
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <asset_dir> <config 1..5> <spp> <out.ppm> [width height [members]]\n", argv[0]);
    return 2;
  }
  const char *asset_dir = argv[1];
  const int config = atoi(argv[2]);
  const uint32_t spp = (uint32_t)atoi(argv[3]);

  /* host side: scene graph -> flat buffers (MeshServer / BLAS / TLAS / binder) */
  mirth_scene *scene = mirth_scene_new();
  MirthConfigInfo info;
  mirt_ctx *build_ctx = NULL;
  const char *device_build = getenv("MIRT_DEVICE_BUILD");
  if (device_build && atoi(device_build) != 0) {
    CHECK(build_ctx, mirt_create(0, &build_ctx));
    mirth_set_blas_builder(scene, build_on_device, build_ctx);
  }
  if (mirth_build_config(scene, config, asset_dir, 0, 0, &info) != 0 || mirth_bind(scene) != 0) {
    fprintf(stderr, "scene: %s\n", mirth_last_error(scene));
    return 1;
  }
  MirtSceneDesc desc;
  mirth_scene_desc(scene, &desc);
  const uint32_t width = argc > 6 ? (uint32_t)atoi(argv[5]) : info.width;
  const uint32_t height = argc > 6 ? (uint32_t)atoi(argv[6]) : info.height;
  const size_t n_pixels = (size_t)width * height;
  MirtSampleSource *sources = malloc(n_pixels * sizeof *sources);
  mirth_make_sample_sources(width, height, info.seed, 2, sources);
  MirtCamera camera;
  mirth_default_camera(&camera);
  camera.dims[0] = (float)width / (float)height;
  camera.dims[1] = 1.0f;

  /* device side */
  MirtParams params;
  mirt_default_params(&params);
  params.max_bounces = info.max_bounces;
  params.shade_mode = info.shade_mode;
  params.nee = info.nee;
  params.seed = info.seed;
  const int members = argc > 7 ? atoi(argv[7]) : 1;
  mirt_ctx *ctx = NULL;
  mirt_group *grp = NULL;
  if (members > 1) {
    /* one process, several GPUs: replicate, split the samples, reduce into member 0 */
    int ordinals[64] = {0};
    const char *list = getenv("MIRT_GROUP_DEVICES");
    for (int i = 0; i < members && i < 64; ++i) {
      ordinals[i] = list ? atoi(list) : 0;
      if (list) {
        while (*list && *list != ',') ++list;
        if (*list == ',') ++list;
        if (!*list) list = NULL;
      }
    }
    CHECK(ctx, mirt_group_create(ordinals, members < 64 ? members : 64, &grp));
#define GCHECK(call)                                                                             \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != 0) {                                                                              \
      fprintf(stderr, "%s: %s (%s)\n", #call, mirt_status_string(rc_), mirt_group_last_error(grp)); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)
    GCHECK(mirt_group_scene_upload(grp, &desc));
    GCHECK(mirt_group_state_create(grp, width, height, (uint32_t)n_pixels, sources, (uint32_t)n_pixels, &params));
    GCHECK(mirt_group_camera_set(grp, &camera));
    GCHECK(mirt_group_render_spp(grp, spp));
    GCHECK(mirt_group_reduce(grp));
    ctx = mirt_group_member(grp, 0); /* the reads below go through member 0 */
  } else {
    CHECK(ctx, mirt_create(0, &ctx));
    CHECK(ctx, mirt_scene_upload(ctx, &desc));
    CHECK(ctx, mirt_state_create(ctx, width, height, (uint32_t)n_pixels, sources, (uint32_t)n_pixels, &params));
    CHECK(ctx, mirt_camera_set(ctx, &camera));
    CHECK(ctx, mirt_render_spp(ctx, spp));
  }

  uint32_t *rgb = malloc(n_pixels * sizeof *rgb);
  uint32_t *sum4 = malloc(4 * n_pixels * sizeof *sum4), *count = malloc(n_pixels * sizeof *count);
  CHECK(ctx, mirt_read_output(ctx, rgb, n_pixels));
  CHECK(ctx, mirt_read_accum(ctx, sum4, count, n_pixels));
  uint64_t checksum = 0, samples = 0;
  for (size_t i = 0; i < n_pixels; ++i) {
    checksum += (uint64_t)sum4[4 * i] + sum4[4 * i + 1] + sum4[4 * i + 2];
    samples += count[i];
  }
  printf("checksum %llu samples %llu\n", (unsigned long long)checksum, (unsigned long long)samples);

  FILE *f = fopen(argv[4], "wb");
  if (!f) {
    perror(argv[4]);
    return 1;
  }
  fprintf(f, "P6\n%u %u\n255\n", width, height);
  for (size_t i = 0; i < n_pixels; ++i) {
    const unsigned char px[3] = {(unsigned char)(rgb[i] & 255u), (unsigned char)((rgb[i] >> 8) & 255u),
                                 (unsigned char)((rgb[i] >> 16) & 255u)};
    fwrite(px, 1, 3, f);
  }
  fclose(f);

  if (grp) mirt_group_destroy(grp);
  else mirt_destroy(ctx);
  if (build_ctx) mirt_destroy(build_ctx);
  mirth_scene_free(scene);
  free(sources);
  free(rgb);
  free(sum4);
  free(count);
  return 0;
}
