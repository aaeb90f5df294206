/*
 * mirt_host.h — C ABI of the host scene model (libmirt_host).
 *
 * The reference's host data model is Rust (MeshServer, MaterialServer,
 * Transform/Instance entities, BLAS/TLAS builders, binder, scenes, camera).
 * This library restates it in C++ so that scenes can be built where no Rust
 * toolchain exists; it produces the flat buffers of MirtSceneDesc that
 * mirt_scene_upload() consumes.  It contains no GPU code and never calls
 * libmirt.
 */
#ifndef MIRT_HOST_H
#define MIRT_HOST_H

#include "mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirth_scene mirth_scene;

mirth_scene *mirth_scene_new(void);
void mirth_scene_free(mirth_scene *s);
const char *mirth_last_error(const mirth_scene *s);

/* MeshServer::load_mesh (src/mesh.rs:149-165).  Each returns a mesh id >= 0
 * or a negative value on failure.  Meshes are de-duplicated by descriptor. */
int mirth_load_mesh_obj(mirth_scene *s, const char *path);   /* MeshDescriptor::TOBJ */
int mirth_load_mesh_rect(mirth_scene *s);                     /* MeshDescriptor::Rect */
int mirth_load_mesh_cube(mirth_scene *s);                     /* MeshDescriptor::Cube */
/* A parsed model (what tobj hands to Mesh::from_model, src/mesh.rs:308):
 * positions/normals are 3 floats each, indices 3 per triangle. */
int mirth_load_mesh_model(mirth_scene *s, const char *name, const float *positions,
                          uint32_t n_positions, const float *normals, uint32_t n_normals,
                          const uint32_t *indices, uint32_t n_indices);
/* Binary dump of a parsed model ("MIRTMESH" file), see mirth_obj_to_model_file. */
int mirth_load_mesh_model_file(mirth_scene *s, const char *path);
/* Procedural dragon-class mesh: 2*n_u*n_v triangles. */
int mirth_load_mesh_knot(mirth_scene *s, uint32_t n_u, uint32_t n_v, uint32_t seed);
/* <asset_dir>/<name>.obj (or .mirtmesh) with every triangle split into four
 * `levels` times and the new vertices jittered by a seeded +-4 % of their edge:
 * a scan-like mesh with the asset's topology (suzanne x 4^5 = 991 232
 * triangles).  File normals are dropped (computed normals, src/mesh.rs:357-413). */
int mirth_load_mesh_subdivided(mirth_scene *s, const char *asset_dir, const char *name, uint32_t levels,
                               uint32_t seed);
/* A texture for MirtMaterial.colour_texture / emissive_texture: width x height
 * RGBA8 texels, MIRT_FILTER_*.  Texels are LINEAR: the sampler reads a channel as
 * value / 255 with no transfer curve, so an sRGB-encoded image is decoded by whoever
 * adds it (mirth_load_gltf does).  Returns its 1-based id (0 in a material means
 * none).  The reference has the ids and no texture store (src/material.rs:15-18). */
int mirth_add_texture(mirth_scene *s, uint32_t width, uint32_t height, const uint32_t *rgba8, uint32_t filter);
/* The same from a PNG image in memory (8 or 16 bits per channel, any colour type, not
 * interlaced; rows top to bottom, so v = 0 is the top row as in glTF). */
int mirth_add_texture_png(mirth_scene *s, const void *png, size_t len, uint32_t filter);

/* What mirth_load_gltf did with the file. */
typedef struct MirthGltfInfo {
  uint32_t n_meshes;              /* meshes added to the MeshServer (a primitive shared by several nodes counts once) */
  uint32_t n_instances;           /* entities spawned */
  uint32_t n_materials;           /* materials added */
  uint32_t n_textures;            /* images decoded into the texture store */
  uint32_t n_baked;               /* instances whose sheared world matrix was applied to a copy of the mesh */
  uint32_t n_skipped_primitives;  /* not TRIANGLES, no POSITION, morph targets, or unreadable accessors */
  uint32_t n_skipped_images;      /* not PNG, interlaced, or unreadable */
  uint32_t reserved;
} MirthGltfInfo;
/* A glTF 2.0 scene (.gltf with external or base64 buffers, or .glb): every triangle primitive the
 * default scene's nodes refer to becomes a mesh (positions as in the file, no normalisation), every
 * reference to it an entity with the node's world transform and the primitive's material, PNG
 * base-colour and emissive textures go to the texture store with their colour channels decoded
 * from sRGB (what glTF defines both as) to the nearest linear 8-bit value.  A sheared world matrix
 * is baked into a copy of the mesh (the file's normals through the inverse transpose).  The node
 * graph must be a tree: a node reached a second time (cycle, diamond) is skipped.  Finishes the loader the reference
 * leaves commented out (src/scenes.rs:332-399).  Returns the number of entities spawned, or -1
 * (mirth_last_error says why).  `info` may be NULL. */
int mirth_load_gltf(mirth_scene *s, const char *path, MirthGltfInfo *info);
/* Vertex uvs of a loaded mesh from two of its normalised position axes:
 * uv = position[axis] * scale + offset.  (The reference uploads uv = 0 for
 * every vertex, src/mesh.rs:241; so does this library until this is called.) */
int mirth_mesh_set_planar_uv(mirth_scene *s, int mesh_id, int axis_u, int axis_v, float scale, float offset);
/* The sin / cos the procedural meshes are built with: IEEE +, -, * on doubles
 * only, so their vertices do not depend on the host's libm. */
void mirth_det_sincos(double x, double *s_out, double *c_out);
/* What mirth_build_config uses when the dragon asset is absent: kind 0 = the
 * knot (default), 1 = suzanne subdivided `levels` times.  Call before
 * mirth_build_config. */
int mirth_set_dragon_standin(mirth_scene *s, int kind, uint32_t levels);

/* The BLAS of every mesh loaded AFTER this call is built by `fn` instead of the
 * host builder (NULL restores the host builder).  `fn` has the signature of
 * mirt_build_blas with a user pointer in place of the context, so a caller
 * that links both libraries can build on the device:
 *   int on_device(void *ctx, const float *p, uint32_t np, uint32_t *f, uint32_t nf,
 *                 MirtBVHNode *nodes, uint32_t cap, uint32_t *n)
 *   { return mirt_build_blas((mirt_ctx *)ctx, p, np, f, nf, nodes, cap, n, NULL); }
 * It must reorder faces4 in place and fill `nodes` exactly as BLAS::new does
 * (src/blas.rs:62-78) and return 0; any other value fails the load. */
typedef int (*mirth_blas_builder_fn)(void *user, const float *positions4, uint32_t n_positions,
                                     uint32_t *faces4, uint32_t n_faces, MirtBVHNode *nodes,
                                     uint32_t nodes_capacity, uint32_t *n_nodes);
int mirth_set_blas_builder(mirth_scene *s, mirth_blas_builder_fn fn, void *user);
/* The host builder on its own, on positions taken as they are (no
 * Mesh::from_model normalisation): BLAS::new (src/blas.rs:62-78).  Same
 * contract as the callback above; *n_nodes holds the count needed when
 * nodes_capacity is too small (the call then returns -1). */
int mirth_build_blas_host(const float *positions4, uint32_t n_positions, uint32_t *faces4,
                          uint32_t n_faces, MirtBVHNode *nodes, uint32_t nodes_capacity,
                          uint32_t *n_nodes);

/* MaterialServer::add_material (src/material.rs:54-57) */
int mirth_add_material(mirth_scene *s, const MirtMaterial *m);
void mirth_default_material(MirtMaterial *out); /* src/material.rs:27-42 */
void mirth_default_camera(MirtCamera *out);     /* src/camera.rs:117-128 */

/* commands.spawn((Transform, MaterialId, MeshId)) */
int mirth_spawn(mirth_scene *s, const MirtTransform *t, int mesh_id, int material_id);
/* spawn_cornell (src/scenes.rs:20-134) */
int mirth_spawn_cornell(mirth_scene *s, const float dims[3], const float pos[3], int with_ceiling_light);

/* One of the five benchmark configurations (1..5).  `asset_dir` is searched
 * for <name>.obj then <name>.mirtmesh for suzanne / teapot / dragon; a missing
 * dragon falls back to the procedural knot.  Writes the image size, samples
 * per pixel, bounce limit, shade mode and whether the dragon was procedural. */
typedef struct MirthConfigInfo {
  uint32_t width, height, spp, max_bounces, shade_mode, nee;
  uint32_t procedural_dragon; /* 0 = asset, 1 = procedural knot, 2 = subdivided suzanne */
  uint32_t seed;
} MirthConfigInfo;
int mirth_build_config(mirth_scene *s, int config, const char *asset_dir, uint32_t dragon_n_u,
                       uint32_t dragon_n_v, MirthConfigInfo *info);

/* regenerate_buffer + binder_system: pack the flat buffers. */
int mirth_bind(mirth_scene *s);
/* Pointers stay valid until the next mirth_bind / mirth_scene_free. */
int mirth_scene_desc(const mirth_scene *s, MirtSceneDesc *out);

/* Sample sources for a width x height image (src/pathtracer_state.rs:175-195),
 * seeded; shuffle 2 = reference order (tiles and sources shuffled), 1 = tiles
 * only, 0 = none.  `out` has width*height entries. */
int mirth_make_sample_sources(uint32_t width, uint32_t height, uint64_t seed, int shuffle,
                              MirtSampleSource *out);

/* Parse an OBJ and write the parsed first model as a MIRTMESH file. */
int mirth_obj_to_model_file(const char *obj_path, const char *out_path);

/* Stand-alone pieces for tests: parse OBJ text; build a BLAS for a mesh id. */
int mirth_parse_obj_counts(const char *text, size_t len, uint32_t *n_positions, uint32_t *n_normals,
                           uint32_t *n_triangles);
int mirth_mesh_counts(const mirth_scene *s, int mesh_id, uint32_t *n_vertices, uint32_t *n_faces,
                      uint32_t *n_nodes);
int mirth_mesh_copy(const mirth_scene *s, int mesh_id, float *positions4, float *normals4,
                    uint32_t *faces4, MirtBVHNode *nodes);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_HOST_H */
