/* fuzz_host_driver.c — feeds one file to a loader of the host library and exits 0 whatever the loader
 * says; a memory error or undefined behaviour is the sanitizer's to report (tools/fuzz_host.py builds the
 * library and this driver with -fsanitize=address,undefined and runs it over mutated inputs).
 *   usage: fuzz_host_driver gltf|png|obj|model <file> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mirt_host.h"

static unsigned char *slurp(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *p = malloc(n > 0 ? (size_t)n : 1);
  *len = p && n > 0 ? fread(p, 1, (size_t)n, f) : 0;
  fclose(f);
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  mirth_scene *s = mirth_scene_new();
  if (!s) return 2;
  int rc = 0;
  if (!strcmp(argv[1], "gltf")) {
    MirthGltfInfo info;
    rc = mirth_load_gltf(s, argv[2], &info);
    if (rc > 0) rc = mirth_bind(s);  /* TLAS, binder packing and the scene description over what was loaded */
    if (rc == 0) {
      MirtSceneDesc d;
      rc = mirth_scene_desc(s, &d);
    }
  } else if (!strcmp(argv[1], "png")) {
    size_t len = 0;
    unsigned char *p = slurp(argv[2], &len);
    if (p) rc = mirth_add_texture_png(s, p, len, 0);
    free(p);
  } else if (!strcmp(argv[1], "obj")) {
    size_t len = 0;
    unsigned char *p = slurp(argv[2], &len);
    uint32_t a = 0, b = 0, c = 0;
    if (p) rc = mirth_parse_obj_counts((const char *)p, len, &a, &b, &c);
    free(p);
    rc = mirth_load_mesh_obj(s, argv[2]);
  } else if (!strcmp(argv[1], "model")) {
    rc = mirth_load_mesh_model_file(s, argv[2]);
  } else {
    return 2;
  }
  if (rc < 0) fprintf(stderr, "refused: %s\n", mirth_last_error(s));
  mirth_scene_free(s);
  return 0;
}
