/*
 * oracle.c — CPU oracle of the wavefront path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker; never part of the product.
 * PARITY UNPINNED by the reference (it has no tests or fixtures); pinned by
 * known-answer values derived from the reference's formulas and by a second
 * restatement of every function of the path, written in numpy float32 from the
 * shader text (tests/test_oracle_kat.py), that agrees with this one bit for bit.
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference checkout).  The reference evaluates its arithmetic through
 * slangc -> SPIR-V -> naga -> a Vulkan driver, which leaves contraction,
 * normalize(), pow() and min/max-of-NaN driver-defined.  This file fixes one
 * definition, the "arithmetic contract", shared with the HIP kernels:
 *   - f32 everywhere; +,-,*,/ and sqrt correctly rounded; no implicit
 *     contraction (built with -ffp-contract=off);
 *   - dot3(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));
 *   - cross(a,b).x = fma(a.y,b.z, -(a.z*b.y)) and cyclic;
 *   - a 3x4 matrix row times (v,1) = fma(m2,v.z, fma(m1,v.y, m0*v.x)) + m3;
 *     times (v,0) the same without the last addition;
 *   - normalize(v) = v * (1 / sqrt(dot3(v,v)));  length(v) = sqrt(dot3(v,v));
 *   - min/max return the non-NaN operand (fminf/fmaxf);
 *   - pow(x,2) = x*x;  pow(x,5) = NaN for x<0, else (x*x)*(x*x)*x;
 *   - everything else is evaluated left to right as written.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NO_PRIM 0x7FFFFFFFu
#define NO_HIT 0xFFFFFFFFu
#define PI_F 3.14159265f /* float.getPi() */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
  return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return scale3(a, inv); }
static inline float pow2f(float x) { return x * x; }
static inline float pow5f(float x) {
  if (x < 0.0f) return NAN;
  float x2 = x * x;
  return (x2 * x2) * x;
}
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }

/* ------------------------------------------------------------------ scene */

typedef struct { float m[3][4]; } mat34;

struct oracle_scene {
  MirtSceneDesc d; /* owned copies */
  mat34 *M;        /* per instance: object -> world */
  mat34 *Mi;       /* per instance: world -> object */
};

static void *dup_mem(const void *p, size_t n) {
  if (!p || !n) return NULL;
  void *q = malloc(n);
  if (q) memcpy(q, p, n);
  return q;
}

/* float4x4 product as Slang's mul() evaluates it: rows times columns, summed
 * left to right (shaders/common.slang:41-111). */
static void mat4_mul(const float a[4][4], const float b[4][4], float o[4][4]) {
  float t[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = a[i][0] * b[0][j];
      acc = acc + a[i][1] * b[1][j];
      acc = acc + a[i][2] * b[2][j];
      acc = acc + a[i][3] * b[3][j];
      t[i][j] = acc;
    }
  memcpy(o, t, sizeof(t));
}
static void mat4_identity(float m[4][4]) {
  memset(m, 0, 64);
  m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1.0f;
}
/* rotate_matrix (shaders/common.slang:50-76) */
static void rotate_matrix(const float r[3], float o[4][4]) {
  float x = r[0], y = r[1], z = r[2];
  float rx[4][4], ry[4][4], rz[4][4], t[4][4];
  mat4_identity(rx); mat4_identity(ry); mat4_identity(rz);
  rx[1][1] = cosf(x); rx[1][2] = -sinf(x); rx[2][1] = sinf(x); rx[2][2] = cosf(x);
  ry[0][0] = cosf(y); ry[0][2] = sinf(y); ry[2][0] = -sinf(y); ry[2][2] = cosf(y);
  rz[0][0] = cosf(z); rz[0][1] = -sinf(z); rz[1][0] = sinf(z); rz[1][1] = cosf(z);
  mat4_mul(ry, rz, t);
  mat4_mul(rx, t, o);
}
/* Transform.matrix() / matrix_inverse() (shaders/common.slang:92-110) */
static void transform_matrices(const MirtTransform *t, mat34 *M, mat34 *Mi) {
  float S[4][4], R[4][4], T[4][4], RS[4][4], m[4][4];
  mat4_identity(S); S[0][0] = t->scale[0]; S[1][1] = t->scale[1]; S[2][2] = t->scale[2];
  rotate_matrix(t->rotation, R);
  mat4_identity(T); T[0][3] = t->translation[0]; T[1][3] = t->translation[1]; T[2][3] = t->translation[2];
  mat4_mul(R, S, RS);
  mat4_mul(T, RS, m);
  memcpy(M->m, m, 48);

  float Si[4][4], Rt[4][4], Ti[4][4], RtTi[4][4], mi[4][4];
  mat4_identity(Si);
  Si[0][0] = 1.0f / t->scale[0]; Si[1][1] = 1.0f / t->scale[1]; Si[2][2] = 1.0f / t->scale[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Rt[i][j] = R[j][i];
  mat4_identity(Ti); Ti[0][3] = -t->translation[0]; Ti[1][3] = -t->translation[1]; Ti[2][3] = -t->translation[2];
  mat4_mul(Rt, Ti, RtTi);
  mat4_mul(Si, RtTi, mi);
  memcpy(Mi->m, mi, 48);
}
static inline v3 xform_point(const mat34 *m, v3 p) {
  return V3(fmaf(m->m[0][2], p.z, fmaf(m->m[0][1], p.y, m->m[0][0] * p.x)) + m->m[0][3],
            fmaf(m->m[1][2], p.z, fmaf(m->m[1][1], p.y, m->m[1][0] * p.x)) + m->m[1][3],
            fmaf(m->m[2][2], p.z, fmaf(m->m[2][1], p.y, m->m[2][0] * p.x)) + m->m[2][3]);
}
static inline v3 xform_vector(const mat34 *m, v3 p) {
  return V3(fmaf(m->m[0][2], p.z, fmaf(m->m[0][1], p.y, m->m[0][0] * p.x)),
            fmaf(m->m[1][2], p.z, fmaf(m->m[1][1], p.y, m->m[1][0] * p.x)),
            fmaf(m->m[2][2], p.z, fmaf(m->m[2][1], p.y, m->m[2][0] * p.x)));
}

static int validate(const MirtSceneDesc *d) {
  if (!d->instances || !d->n_instances || !d->geometry_offsets || !d->vertices || !d->indices ||
      !d->blas_nodes || !d->materials || !d->transforms || !d->tlas_nodes || !d->tlas_to_instances)
    return 0;
  for (uint32_t i = 0; i < d->n_instances; ++i)
    if (d->instances[i].transform >= d->n_transforms || d->instances[i].geometry >= d->n_geometries ||
        d->instances[i].material >= d->n_materials)
      return 0;
  for (uint32_t i = 0; i < d->n_tlas_to_instances; ++i)
    if (d->tlas_to_instances[i] >= d->n_instances) return 0;
  for (uint32_t i = 0; i < d->n_tlas_nodes; ++i) {
    const MirtBVHNode *n = &d->tlas_nodes[i];
    if (n->right >= d->n_tlas_nodes) return 0;
    if (n->is_leaf ? (n->end > d->n_tlas_to_instances || n->start > n->end) : (n->left >= d->n_tlas_nodes)) return 0;
  }
  for (uint32_t g = 0; g < d->n_geometries; ++g) {
    const MirtGeometryOffsets *o = &d->geometry_offsets[g];
    uint32_t node_end = g + 1 < d->n_geometries ? d->geometry_offsets[g + 1].blas_node : d->n_blas_nodes;
    uint32_t idx_end = g + 1 < d->n_geometries ? d->geometry_offsets[g + 1].index : d->n_indices;
    uint32_t vtx_end = g + 1 < d->n_geometries ? d->geometry_offsets[g + 1].vertex : d->n_vertices;
    if (o->blas_node >= node_end || node_end > d->n_blas_nodes || idx_end > d->n_indices || vtx_end > d->n_vertices)
      return 0;
    uint32_t nn = node_end - o->blas_node, ni = idx_end - o->index, nv = vtx_end - o->vertex;
    for (uint32_t i = 0; i < nn; ++i) {
      const MirtBVHNode *n = &d->blas_nodes[o->blas_node + i];
      if (n->right >= nn) return 0;
      if (n->is_leaf ? (n->end > ni || n->start > n->end) : (n->left >= nn)) return 0;
    }
    for (uint32_t i = 0; i < ni; ++i)
      for (int k = 0; k < 3; ++k)
        if (d->indices[o->index + i].v[k] >= nv) return 0;
  }
  for (uint32_t i = 0; i < d->n_light_sources; ++i)
    if (d->light_sources[i] != NO_HIT && d->light_sources[i] >= d->n_instances) return 0;
  if (d->n_classic_refs) {
    if (d->n_classic_refs != d->n_materials) return 0;
    for (uint32_t i = 0; i < d->n_classic_refs; ++i) {
      const MirtClassicMaterialRef r = d->classic_refs[i];
      uint32_t lim = r.kind == MIRT_MAT_LAMBERTIAN ? d->n_lambertian
                   : r.kind == MIRT_MAT_METALLIC   ? d->n_metallic
                   : r.kind == MIRT_MAT_DIELECTRIC ? d->n_dielectric
                   : r.kind == MIRT_MAT_EMISSIVE   ? d->n_emissive : 0xFFFFFFFFu;
      if (r.kind > MIRT_MAT_EMISSIVE || (r.kind != MIRT_MAT_NONE && r.data >= lim)) return 0;
    }
  }
  if (d->n_textures && (!d->textures || !d->texels)) return 0;
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const MirtTexture *t = &d->textures[i];
    if (!t->width || !t->height || t->filter > MIRT_FILTER_BILINEAR ||
        (uint64_t)t->first_texel + (uint64_t)t->width * t->height > d->n_texels)
      return 0;
  }
  for (uint32_t i = 0; i < d->n_materials; ++i)
    if (d->materials[i].colour_texture > d->n_textures || d->materials[i].emissive_texture > d->n_textures) return 0;
  return 1;
}

oracle_scene *oracle_scene_create(const MirtSceneDesc *desc) {
  if (!desc || !validate(desc)) return NULL;
  oracle_scene *s = (oracle_scene *)calloc(1, sizeof(*s));
  if (!s) return NULL;
  s->d = *desc;
#define DUP(field, count, type) s->d.field = (const type *)dup_mem(desc->field, (size_t)desc->count * sizeof(type))
  DUP(instances, n_instances, MirtInstance);
  DUP(geometry_offsets, n_geometries, MirtGeometryOffsets);
  DUP(vertices, n_vertices, MirtVertex);
  DUP(indices, n_indices, MirtIndex);
  DUP(blas_nodes, n_blas_nodes, MirtBVHNode);
  DUP(materials, n_materials, MirtMaterial);
  DUP(transforms, n_transforms, MirtTransform);
  DUP(tlas_nodes, n_tlas_nodes, MirtBVHNode);
  DUP(tlas_to_instances, n_tlas_to_instances, uint32_t);
  DUP(light_sources, n_light_sources, uint32_t);
  DUP(classic_refs, n_classic_refs, MirtClassicMaterialRef);
  DUP(lambertian, n_lambertian, MirtLambertianData);
  DUP(metallic, n_metallic, MirtMetallicData);
  DUP(dielectric, n_dielectric, MirtDielectricData);
  DUP(emissive, n_emissive, MirtEmissiveData);
  DUP(textures, n_textures, MirtTexture);
  DUP(texels, n_texels, uint32_t);
#undef DUP
  s->M = (mat34 *)malloc(sizeof(mat34) * desc->n_instances);
  s->Mi = (mat34 *)malloc(sizeof(mat34) * desc->n_instances);
  for (uint32_t i = 0; i < desc->n_instances; ++i)
    transform_matrices(&s->d.transforms[s->d.instances[i].transform], &s->M[i], &s->Mi[i]);
  return s;
}

void oracle_scene_free(oracle_scene *s) {
  if (!s) return;
  free((void *)s->d.instances); free((void *)s->d.geometry_offsets); free((void *)s->d.vertices);
  free((void *)s->d.indices); free((void *)s->d.blas_nodes); free((void *)s->d.materials);
  free((void *)s->d.transforms); free((void *)s->d.tlas_nodes); free((void *)s->d.tlas_to_instances);
  free((void *)s->d.light_sources); free((void *)s->d.classic_refs); free((void *)s->d.lambertian);
  free((void *)s->d.metallic); free((void *)s->d.dielectric); free((void *)s->d.emissive);
  free((void *)s->d.textures); free((void *)s->d.texels);
  free(s->M); free(s->Mi); free(s);
}

/* -------------------------------------------------------------------- RNG */

/* TausStep / LCGStep / Random (shaders/random.slang:9-30) */
static inline uint32_t taus_step(uint32_t z, int s1, int s2, int s3, uint32_t m) {
  uint32_t b = (((z << s1) ^ z) >> s2);
  return (((z & m) << s3) ^ b);
}
float oracle_rng_next(uint32_t st[4]) {
  st[0] = taus_step(st[0], 13, 19, 12, 4294967294u);
  st[1] = taus_step(st[1], 2, 25, 4, 4294967288u);
  st[2] = taus_step(st[2], 3, 11, 17, 4294967280u);
  st[3] = 1664525u * st[3] + 1013904223u;
  return 2.3283064365387e-10f * (float)(st[0] ^ st[1] ^ st[2] ^ st[3]);
}

/* The reference seeds one persistent stream per path slot from OS entropy,
 * each word in [1000, 2^32) (src/pathtracer_state.rs:101-110), which makes a
 * sample's stream depend on which slot picked it up.  Here a stream is keyed
 * by (seed, source index, sample number) instead, so that any slot, any queue
 * order and any GPU produce the same sample. */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
void oracle_rng_init(uint32_t seed, uint32_t source, uint32_t sample_no, uint32_t st[4]) {
  for (uint32_t i = 0; i < 4; ++i) {
    uint32_t w = mix32(mix32(mix32(seed + 0x9E3779B9u * (i + 1u)) ^ source) ^ sample_no);
    if (w < 1000u) w += 1000u;
    st[i] = w;
  }
}

/* -------------------------------------------------------------- traversal */

typedef struct {
  float t, u, v;
  uint32_t prim, inst;
  int found;
} besthit;

typedef struct { uint64_t tlas_nodes, instance_visits, blas_nodes, triangles; } tcounts;

/* rayBoxIntersect (shaders/ray_extend.slang:94-111); dir_inv hoisted. */
static inline int ray_box(v3 pos, v3 dir_inv, const float *lb, const float *ub, float *tmin_out, float *tmax_io) {
  float tmin = -FLT_MAX; /* float.minValue = lowest finite float */
  float tmax = *tmax_io;
  const float p[3] = {pos.x, pos.y, pos.z}, di[3] = {dir_inv.x, dir_inv.y, dir_inv.z};
  for (int d = 0; d < 3; ++d) {
    int sign = di[d] >= 0.0f;
    float bmin = sign ? lb[d] : ub[d];
    float bmax = sign ? ub[d] : lb[d];
    float dmin = (bmin - p[d]) * di[d];
    float dmax = (bmax - p[d]) * di[d];
    tmin = fmaxf(dmin, tmin);
    tmax = fminf(dmax, tmax);
  }
  *tmin_out = tmin;
  *tmax_io = tmax;
  return tmin <= tmax;
}

/* rayTriIntersect (shaders/ray_extend.slang:53-92), barycentrics only; the
 * interpolated vertex is rebuilt for the winning triangle afterwards. */
static inline int ray_tri(v3 pos, v3 dir, v3 p0, v3 p1, v3 p2, float *t, float *u_out, float *v_out) {
  v3 e1 = sub3(p1, p0), e2 = sub3(p2, p0);
  v3 q = cross3(dir, e2);
  float alpha = dot3(e1, q);
  if (alpha > -(10e-8f) && alpha < 10e-8f) return 0;
  float f = 1.0f / alpha;
  v3 s = sub3(pos, p0);
  float u = f * dot3(s, q);
  if (u < 0.0f) return 0;
  v3 r = cross3(s, e1);
  float v = f * dot3(dir, r);
  if (v < 0.0f || u + v > 1.0f) return 0;
  float t2 = f * dot3(e2, r);
  if (t2 > *t || t2 < 0.0f) return 0;
  *t = t2;
  *u_out = u;
  *v_out = v;
  return 1;
}

/* blasFirstHit (shaders/ray_extend.slang:113-160).  any_hit != 0 returns at
 * the first accepted triangle (shadow / connect query). */
static int blas_first_hit(const oracle_scene *s, v3 pos, v3 dir, uint32_t instance_id, uint32_t last_inst,
                          uint32_t last_prim, float *t, besthit *h, tcounts *c, int any_hit) {
  const MirtInstance inst = s->d.instances[instance_id];
  const MirtGeometryOffsets off = s->d.geometry_offsets[inst.geometry];
  const v3 dir_inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  uint32_t current = 0;
  int success = 0;
  do {
    const MirtBVHNode *node = &s->d.blas_nodes[current + off.blas_node];
    c->blas_nodes++;
    float tmax_aabb = *t, tmin_aabb;
    int hit_aabb = ray_box(pos, dir_inv, node->lb, node->ub, &tmin_aabb, &tmax_aabb);
    current = hit_aabb ? node->left : node->right;
    current = node->is_leaf == 1 ? node->right : current;
    if (!hit_aabb || node->is_leaf == 0) continue;
    for (uint32_t p = node->start; p < node->end; ++p) {
      if (!(p == last_prim && instance_id == last_inst)) {
        const MirtIndex face = s->d.indices[p + off.index];
        const v3 p0 = ld3(s->d.vertices[face.v[0] + off.vertex].position);
        const v3 p1 = ld3(s->d.vertices[face.v[1] + off.vertex].position);
        const v3 p2 = ld3(s->d.vertices[face.v[2] + off.vertex].position);
        c->triangles++;
        float t2 = *t, u, v;
        if (ray_tri(pos, dir, p0, p1, p2, &t2, &u, &v)) {
          *t = t2;
          h->t = t2; h->u = u; h->v = v; h->prim = p;
          success = 1;
          if (any_hit) return 1;
        }
      }
    }
  } while (current != 0);
  return success;
}

/* tlasFirstHit (shaders/ray_extend.slang:162-215) */
static int tlas_first_hit(const oracle_scene *s, v3 pos, v3 dir, uint32_t last_inst, uint32_t last_prim,
                          float *t, besthit *h, tcounts *c, int any_hit) {
  const v3 dir_inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  uint32_t current = 0;
  int success = 0;
  do {
    const MirtBVHNode *node = &s->d.tlas_nodes[current];
    c->tlas_nodes++;
    float tmax_aabb = *t, tmin_aabb;
    int hit_aabb = ray_box(pos, dir_inv, node->lb, node->ub, &tmin_aabb, &tmax_aabb) &&
                   (tmax_aabb >= 0.0f || tmin_aabb >= 0.0f);
    current = hit_aabb ? node->left : node->right;
    current = node->is_leaf == 1 ? node->right : current;
    if (!hit_aabb || node->is_leaf == 0) continue;
    for (uint32_t i = node->start; i < node->end; ++i) {
      const uint32_t id = s->d.tlas_to_instances[i];
      c->instance_visits++;
      const v3 rpos = xform_point(&s->Mi[id], pos);
      const v3 rdir = xform_vector(&s->Mi[id], dir);
      float t2 = *t;
      besthit h2;
      if (blas_first_hit(s, rpos, rdir, id, last_inst, last_prim, &t2, &h2, c, any_hit)) {
        *t = t2;
        *h = h2;
        h->inst = id;
        success = 1;
        if (any_hit) return 1;
      }
    }
  } while (current != 0);
  return success;
}

/* The hit record extensionMain stores (shaders/ray_extend.slang:86-89,
 * 204-207): interpolated position and normal moved to world space. */
typedef struct { v3 position, normal; uint32_t prim, inst, front_face; float tu, tv; } hitrec;

static void finish_hit(const oracle_scene *s, v3 world_dir, const besthit *b, hitrec *out) {
  const MirtInstance inst = s->d.instances[b->inst];
  const MirtGeometryOffsets off = s->d.geometry_offsets[inst.geometry];
  const MirtIndex face = s->d.indices[b->prim + off.index];
  const MirtVertex *v0 = &s->d.vertices[face.v[0] + off.vertex];
  const MirtVertex *v1 = &s->d.vertices[face.v[1] + off.vertex];
  const MirtVertex *v2 = &s->d.vertices[face.v[2] + off.vertex];
  const v3 p0 = ld3(v0->position), e1 = sub3(ld3(v1->position), p0), e2 = sub3(ld3(v2->position), p0);
  const v3 n0 = ld3(v0->normal), n1 = ld3(v1->normal), n2 = ld3(v2->normal);
  const float u = b->u, v = b->v, w = 1.0f - u - v;
  const v3 n = V3(fmaf(n2.x, v, fmaf(n1.x, u, n0.x * w)), fmaf(n2.y, v, fmaf(n1.y, u, n0.y * w)),
                  fmaf(n2.z, v, fmaf(n1.z, u, n0.z * w)));
  const v3 p = V3(fmaf(e2.x, v, fmaf(e1.x, u, p0.x)), fmaf(e2.y, v, fmaf(e1.y, u, p0.y)),
                  fmaf(e2.z, v, fmaf(e1.z, u, p0.z)));
  out->position = xform_point(&s->M[b->inst], p);
  out->normal = normalize3(xform_vector(&s->M[b->inst], n)); /* M, not its inverse transpose */
  out->front_face = dot3(out->normal, world_dir) < 0.0f;
  out->prim = b->prim;
  out->inst = b->inst;
  /* texture coordinate: the vertex uvs interpolated like the normal */
  out->tu = fmaf(v2->uv[0], v, fmaf(v1->uv[0], u, v0->uv[0] * w));
  out->tv = fmaf(v2->uv[1], v, fmaf(v1->uv[1], u, v0->uv[1] * w));
}

/* ---------------------------------------------------------------- textures
 * The reference carries texture ids in Material and binds no texture
 * (src/material.rs:15-18), so the sampling rule is the build's own (mirt.h):
 * RGBA8, channel / 255, repeat addressing, nearest or bilinear with texel
 * centres at (i + 0.5) / size; id 0 = none, id t = textures[t - 1]. */
static v3 texel_rgb(uint32_t t) {
  return V3((float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f);
}
static v3 tex_sample(const oracle_scene *s, uint32_t id, float tu, float tv) {
  const MirtTexture t = s->d.textures[id - 1u];
  const uint32_t W = t.width, H = t.height;
  const uint32_t *px = s->d.texels + t.first_texel;
  const float fu = tu - floorf(tu), fv = tv - floorf(tv);
  if (t.filter == MIRT_FILTER_NEAREST) {
    const float xf = fu * (float)W, yf = fv * (float)H;
    uint32_t x = xf >= 0.0f ? (uint32_t)xf : 0u, y = yf >= 0.0f ? (uint32_t)yf : 0u; /* NaN: texel 0 */
    if (x >= W) x = W - 1u;
    if (y >= H) y = H - 1u;
    return texel_rgb(px[(size_t)y * W + x]);
  }
  const float sx = fu * (float)W - 0.5f, sy = fv * (float)H - 0.5f;
  const float x0f = floorf(sx), y0f = floorf(sy);
  const float tx = sx - x0f, ty = sy - y0f;
  const int xi = (x0f >= -1.0f) ? (int)x0f : 0, yi = (y0f >= -1.0f) ? (int)y0f : 0;
  const uint32_t x0 = xi < 0 ? W - 1u : ((uint32_t)xi >= W ? W - 1u : (uint32_t)xi);
  const uint32_t y0 = yi < 0 ? H - 1u : ((uint32_t)yi >= H ? H - 1u : (uint32_t)yi);
  const uint32_t x1 = x0 + 1u == W ? 0u : x0 + 1u, y1 = y0 + 1u == H ? 0u : y0 + 1u;
  const v3 c00 = texel_rgb(px[(size_t)y0 * W + x0]), c10 = texel_rgb(px[(size_t)y0 * W + x1]);
  const v3 c01 = texel_rgb(px[(size_t)y1 * W + x0]), c11 = texel_rgb(px[(size_t)y1 * W + x1]);
  const float ax = 1.0f - tx, ay = 1.0f - ty;
  const v3 top = V3(c00.x * ax + c10.x * tx, c00.y * ax + c10.y * tx, c00.z * ax + c10.z * tx);
  const v3 bot = V3(c01.x * ax + c11.x * tx, c01.y * ax + c11.y * tx, c01.z * ax + c11.z * tx);
  return V3(top.x * ay + bot.x * ty, top.y * ay + bot.y * ty, top.z * ay + bot.z * ty);
}
/* 1,1,1 or the texel the hit sees through texture `id` */
static v3 hit_texel(const oracle_scene *s, uint32_t id, const hitrec *h) {
  return id != 0u && id <= s->d.n_textures ? tex_sample(s, id, h->tu, h->tv) : V3(1.0f, 1.0f, 1.0f);
}

static void add_counts(MirtTraversalCounts *dst, const tcounts *c, uint64_t rays) {
  if (!dst) return;
  dst->rays += rays;
  dst->tlas_nodes += c->tlas_nodes;
  dst->instance_visits += c->instance_visits;
  dst->blas_nodes += c->blas_nodes;
  dst->triangles += c->triangles;
}

void oracle_trace_closest(const oracle_scene *s, const MirtRay *rays, size_t n, MirtHitId *out,
                          MirtTraversalCounts *counts) {
  tcounts total = {0, 0, 0, 0};
#pragma omp parallel
  {
    tcounts c = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 256)
    for (long i = 0; i < (long)n; ++i) {
      const v3 pos = ld3(rays[i].pos), dir = ld3(rays[i].dir);
      float t = FLT_MAX;
      besthit b;
      MirtHitId o = {NO_HIT, NO_HIT, FLT_MAX, 0};
      if (tlas_first_hit(s, pos, dir, rays[i].last_inst, rays[i].last_prim, &t, &b, &c, 0)) {
        hitrec h;
        finish_hit(s, dir, &b, &h);
        o.instance_id = b.inst; o.triangle_id = b.prim; o.t = t; o.front_face = h.front_face;
      }
      out[i] = o;
    }
#pragma omp critical
    {
      total.tlas_nodes += c.tlas_nodes; total.instance_visits += c.instance_visits;
      total.blas_nodes += c.blas_nodes; total.triangles += c.triangles;
    }
  }
  add_counts(counts, &total, n);
}

void oracle_trace_any(const oracle_scene *s, const MirtRay *rays, const float *tmax, size_t n,
                      MirtHitId *out, MirtTraversalCounts *counts) {
  tcounts total = {0, 0, 0, 0};
#pragma omp parallel
  {
    tcounts c = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 256)
    for (long i = 0; i < (long)n; ++i) {
      const v3 pos = ld3(rays[i].pos), dir = ld3(rays[i].dir);
      float t = tmax[i];
      besthit b;
      MirtHitId o = {NO_HIT, NO_HIT, FLT_MAX, 0};
      if (tlas_first_hit(s, pos, dir, rays[i].last_inst, rays[i].last_prim, &t, &b, &c, 1)) {
        o.instance_id = b.inst; o.triangle_id = b.prim; o.t = t;
      }
      out[i] = o;
    }
#pragma omp critical
    {
      total.tlas_nodes += c.tlas_nodes; total.instance_visits += c.instance_visits;
      total.blas_nodes += c.blas_nodes; total.triangles += c.triangles;
    }
  }
  add_counts(counts, &total, n);
}

void oracle_trace_brute(const oracle_scene *s, const MirtRay *rays, size_t n, MirtHitId *out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < (long)n; ++i) {
    const v3 pos = ld3(rays[i].pos), dir = ld3(rays[i].dir);
    float t = FLT_MAX;
    MirtHitId o = {NO_HIT, NO_HIT, FLT_MAX, 0};
    /* instances in TLAS slot order, triangles in index order: the order the
     * threaded traversal visits them, so that equal-t ties resolve alike */
    for (uint32_t k = 0; k < s->d.n_tlas_to_instances; ++k) {
      const uint32_t id = s->d.tlas_to_instances[k];
      const MirtGeometryOffsets off = s->d.geometry_offsets[s->d.instances[id].geometry];
      const uint32_t g = s->d.instances[id].geometry;
      const uint32_t idx_end = g + 1 < s->d.n_geometries ? s->d.geometry_offsets[g + 1].index : s->d.n_indices;
      const v3 rpos = xform_point(&s->Mi[id], pos), rdir = xform_vector(&s->Mi[id], dir);
      for (uint32_t p = 0; p < idx_end - off.index; ++p) {
        if (p == rays[i].last_prim && id == rays[i].last_inst) continue;
        const MirtIndex face = s->d.indices[p + off.index];
        float u, v;
        if (ray_tri(rpos, rdir, ld3(s->d.vertices[face.v[0] + off.vertex].position),
                    ld3(s->d.vertices[face.v[1] + off.vertex].position),
                    ld3(s->d.vertices[face.v[2] + off.vertex].position), &t, &u, &v)) {
          o.instance_id = id; o.triangle_id = p; o.t = t;
        }
      }
    }
    out[i] = o;
  }
}

/* ------------------------------------------------------------ path state */

typedef struct {
  v3 rad, throughput;
  uint32_t bounces, source, sample_no;
  uint32_t rng[4];
  v3 ray_pos, ray_dir;
  uint32_t last_inst, last_prim;
  hitrec hit;
  /* the connect (shadow) ray traced at the latest hit, if any */
  MirtRay connect_ray;
  float connect_tmax;
  int connect_valid;
} path;

/* spawnSample (shaders/sample.slang:59-102) */
static void spawn_sample(path *p, const MirtCamera *cam, const MirtSampleSource *src, uint32_t source,
                         uint32_t sample_no, uint32_t width, uint32_t height, const MirtParams *prm) {
  p->bounces = prm->max_bounces;
  p->rad = V3(0, 0, 0);
  p->source = source;
  p->sample_no = sample_no;
  p->throughput = V3(1.0f, 1.0f, 1.0f);
  /* per-slot streams (shaders/random.slang:32-36) are never re-seeded */
  if (prm->rng_mode != MIRT_RNG_PER_SLOT) oracle_rng_init(prm->seed, source, sample_no, p->rng);
  p->ray_pos = ld3(cam->position);
  p->last_inst = NO_PRIM; /* int.maxValue */
  p->last_prim = NO_PRIM;
  const v3 fwd = ld3(cam->forward), up = ld3(cam->up);
  const v3 right = cross3(fwd, up);
  const float dx = cam->dims[0], dy = cam->dims[1];
  const v3 top_left = sub3(add3(scale3(fwd, cam->focal_length), scale3(up, dy)), scale3(right, dx));
  const float r1 = oracle_rng_next(p->rng); /* x jitter takes the first draw */
  const float r2 = oracle_rng_next(p->rng);
  const float jx = r1 / (float)width, jy = r2 / (float)height;
  const float sy = src->screen_pos[1] + jy, sx = src->screen_pos[0] + jx;
  const v3 offset = add3(scale3(scale3(scale3(up, -2.0f), dy), sy), scale3(scale3(scale3(right, 2.0f), dx), sx));
  p->ray_dir = normalize3(add3(top_left, offset));
}

/* unitDiskSample (shaders/shade.slang:101-109) */
static void unit_disk_sample(uint32_t rng[4], float *x, float *y) {
  float px, py;
  do {
    px = 2.0f * oracle_rng_next(rng) - 1.0f;
    py = 2.0f * oracle_rng_next(rng) - 1.0f;
  } while (sqrtf(fmaf(py, py, px * px)) > 1.0f);
  *x = px;
  *y = py;
}
/* unitSphereSample (shaders/shade.slang:115-124): positive octant only; only
 * its draw count is observable (its value feeds dead code in shadeMain). */
static void unit_sphere_sample_draws(uint32_t rng[4]) {
  v3 p;
  do {
    float r1 = oracle_rng_next(rng), r2 = oracle_rng_next(rng), r3 = oracle_rng_next(rng);
    p = V3(r1, r2, r3);
  } while (length3(p) > 1.0f);
}
/* cosineHemisphereSample (shaders/shade.slang:130-140) */
static v3 cosine_hemisphere_sample(v3 n, uint32_t rng[4]) {
  float dx, dy;
  unit_disk_sample(rng, &dx, &dy);
  float z = sqrtf(fmaxf(0.0f, 1.0f - dx * dx - dy * dy));
  v3 temp = fabsf(n.x) > 0.9f ? V3(0, 1, 0) : V3(1, 0, 0);
  v3 t1 = normalize3(cross3(n, temp));
  v3 t2 = cross3(n, t1);
  v3 nn = normalize3(n);
  return V3(fmaf(nn.x, z, fmaf(t2.x, dy, t1.x * dx)), fmaf(nn.y, z, fmaf(t2.y, dy, t1.y * dx)),
            fmaf(nn.z, z, fmaf(t2.z, dy, t1.z * dx)));
}

static inline float heaviside(float x) { return x > 0.0f ? 1.0f : 0.0f; }
static inline v3 mix3(v3 a, v3 b, float m) { /* shaders/shade.slang:24-26 */
  float k = 1.0f - m;
  return V3(k * a.x + m * b.x, k * a.y + m * b.y, k * a.z + m * b.z);
}
/* specularBRDF (shaders/shade.slang:40-53) */
static float specular_brdf(v3 wi, v3 wo, v3 n, float alpha) {
  v3 h = normalize3(add3(wo, wi));
  float a2 = alpha * alpha;
  float nh = dot3(n, h);
  float d = a2 * heaviside(nh);
  d /= PI_F * pow2f(pow2f(nh) * (a2 - 1.0f) + 1.0f);
  float nwi = dot3(n, wi), nwo = dot3(n, wo);
  float nwi2 = pow2f(nwi), nwo2 = pow2f(nwo);
  float v = heaviside(dot3(h, wi)) / (fabsf(nwi) + sqrtf(a2 + (1.0f - a2) * nwi2));
  v *= heaviside(dot3(h, wo)) / (fabsf(nwo) + sqrtf(a2 + (1.0f - a2) * nwo2));
  return v * d;
}
/* specularBTDF (shaders/shade.slang:59-72); all three lanes are equal */
static float specular_btdf(v3 wi, v3 wo, v3 n, float alpha) {
  float nwi = dot3(n, wi), nwo = dot3(n, wo);
  v3 ht = normalize3(add3(sub3(wo, scale3(n, 2.0f * nwi)), wi));
  float a2 = alpha * alpha;
  float nht = dot3(n, ht);
  float dt = a2 * heaviside(nht);
  dt /= PI_F * pow2f(pow2f(nht) * (a2 - 1.0f) + 1.0f);
  float vt = heaviside(dot3(ht, wi) / nwi);
  vt *= heaviside(dot3(ht, wo) / nwo);
  vt /= fabsf(nwi) + sqrtf(a2 + (1.0f - a2)) * pow2f(nwi);
  vt /= fabsf(nwo) + sqrtf(a2 + (1.0f - a2)) * pow2f(nwo);
  return vt * dt;
}
/* material() = mix(dielectricBRDF, metallicBRDF, metallic) (shaders/shade.slang:19-99) */
static v3 material_brdf(v3 wi, v3 wo, v3 n, const MirtMaterial *m) {
  const v3 colour = ld3(m->colour);
  const float alpha = pow2f(m->roughness);
  const v3 h = normalize3(add3(wo, wi));
  const float fres = pow5f(1.0f - fabsf(dot3(wo, h)));
  const float spec = specular_brdf(wi, wo, n, alpha);
  /* dielectricBRDF -> fresnelMix */
  const v3 diffuse = scale3(colour, 1.0f / PI_F);
  const float btdf = specular_btdf(wi, wo, n, alpha);
  const v3 base = mix3(diffuse, scale3(colour, btdf), m->transmission);
  const float f0 = pow2f((1.0f - m->ior) / (1.0f + m->ior));
  const float fr = f0 + (1.0f - f0) * fres;
  const v3 dielectric = mix3(base, V3(spec, spec, spec), fr);
  /* metallicBRDF -> conductorFresnel */
  const v3 metallic = V3(spec * (colour.x + (1.0f - colour.x) * fres), spec * (colour.y + (1.0f - colour.y) * fres),
                         spec * (colour.z + (1.0f - colour.z) * fres));
  return mix3(dielectric, metallic, m->metallic);
}

/* shadeMain (shaders/shade.slang:154-224).  Returns 1 when the path goes back
 * to the extension queue, 0 when it terminates. */
static int shade_uber(const oracle_scene *s, path *p) {
  const v3 wo = p->ray_dir; /* the incoming direction, not negated */
  const MirtInstance inst = s->d.instances[p->hit.inst];
  MirtMaterial textured = s->d.materials[inst.material];
  if (textured.colour_texture != 0u) { /* base-colour texture x colour factor */
    const v3 c = hit_texel(s, textured.colour_texture, &p->hit);
    textured.colour[0] = textured.colour[0] * c.x; textured.colour[1] = textured.colour[1] * c.y;
    textured.colour[2] = textured.colour[2] * c.z;
  }
  if (textured.emissive_texture != 0u) {
    const v3 c = hit_texel(s, textured.emissive_texture, &p->hit);
    textured.emissive[0] = textured.emissive[0] * c.x; textured.emissive[1] = textured.emissive[1] * c.y;
    textured.emissive[2] = textured.emissive[2] * c.z;
  }
  const MirtMaterial *mat = &textured;
  p->rad = add3(p->rad, mul3(p->throughput, ld3(mat->emissive)));
  v3 n = scale3(p->hit.normal, p->hit.front_face != 0 ? 1.0f : -1.0f);
  (void)cosine_hemisphere_sample(n, p->rng); /* diffuse_sample: value unused, draws kept */
  unit_sphere_sample_draws(p->rng);          /* metallicSample -> unitSphereSample */
  const v3 wi = cosine_hemisphere_sample(n, p->rng);
  const float pdf = fabsf(dot3(wi, n)) / PI_F;
  p->ray_dir = wi;
  p->ray_pos = p->hit.position; /* no epsilon: the last primitive is excluded instead */
  p->last_inst = p->hit.inst;
  p->last_prim = p->hit.prim;
  const v3 brdf = material_brdf(wi, wo, n, mat);
  const float ndotwi = fabsf(dot3(n, wi));
  const v3 f = V3(brdf.x * ndotwi * 1.0f / pdf, brdf.y * ndotwi * 1.0f / pdf, brdf.z * ndotwi * 1.0f / pdf);
  p->throughput = mul3(p->throughput, f);
  p->bounces -= 1;
  return p->bounces != 0;
}

/* extensionMain (shaders/ray_extend.slang:19-51).  Returns 1 on a hit. */
static int extend_path(const oracle_scene *s, path *p, const MirtParams *prm, tcounts *c) {
  float t = FLT_MAX;
  besthit b;
  if (!tlas_first_hit(s, p->ray_pos, p->ray_dir, p->last_inst, p->last_prim, &t, &b, c, 0)) {
    const float sky = prm->sky_radiance;
    p->rad = add3(p->rad, mul3(p->throughput, V3(sky, sky, sky)));
    return 0;
  }
  finish_hit(s, p->ray_dir, &b, &p->hit);
  return 1;
}

/* accumulateSample (shaders/sample.slang:35-47).  uint() of a negative or
 * huge float is undefined in the reference; here it clamps to [0, 2^32). */
static inline uint32_t to_fixed(float rad, float scale) {
  if (isnan(rad) || isinf(rad)) rad = 0.0f;
  float x = rad * scale;
  if (!(x >= 0.0f)) return 0u;
  if (x >= 4294967040.0f) return 4294967040u;
  return (uint32_t)x;
}
/* The reference's InterlockedAdd wraps (shaders/sample.slang:45-47); here a sum
 * that would pass 2^32 - 1 stays there, so that the result does not depend on
 * the order of the adds, and the event is counted.  Returns 1 on saturation. */
static int sat_add(uint32_t *p, uint32_t v) {
  uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED), nv;
  int sat;
  do {
    sat = old + v < old;
    nv = sat ? 0xFFFFFFFFu : old + v;
  } while (!__atomic_compare_exchange_n(p, &old, nv, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return sat;
}
/* `moments` (may be NULL): per source the sum and the sum of squares of
 * y = min(r + g + b in fixed point, 2^24 - 1) — the integer form of the
 * reference's declared-but-unused sample_std (shaders/pathtracer.slang:31). */
static uint64_t *g_moments = NULL; /* set for the duration of a render call that keeps moments */
static uint32_t accumulate_sample(const path *p, const MirtParams *prm, uint32_t *sum4, uint32_t *count) {
  const uint32_t fx = to_fixed(p->rad.x, prm->fixed_point_scale);
  const uint32_t fy = to_fixed(p->rad.y, prm->fixed_point_scale);
  const uint32_t fz = to_fixed(p->rad.z, prm->fixed_point_scale);
  if (g_moments) {
    uint64_t y = (uint64_t)fx + (uint64_t)fy + (uint64_t)fz;
    if (y > 0xFFFFFFull) y = 0xFFFFFFull;
    __atomic_fetch_add(&g_moments[2 * (size_t)p->source], y, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_moments[2 * (size_t)p->source + 1], y * y, __ATOMIC_RELAXED);
  }
#pragma omp atomic
  count[p->source] += 1;
  uint32_t sat = 0;
  sat += (uint32_t)sat_add(&sum4[4 * (size_t)p->source + 0], fx);
  sat += (uint32_t)sat_add(&sum4[4 * (size_t)p->source + 1], fy);
  sat += (uint32_t)sat_add(&sum4[4 * (size_t)p->source + 2], fz);
  return sat;
}

static void stats_add(OracleRenderStats *dst, const OracleRenderStats *src) {
  dst->extend_rays += src->extend_rays; dst->connect_rays += src->connect_rays;
  dst->shade_events += src->shade_events; dst->finished_samples += src->finished_samples;
  dst->overflowed_sums += src->overflowed_sums;
  dst->extend_counts.rays += src->extend_counts.rays;
  dst->extend_counts.tlas_nodes += src->extend_counts.tlas_nodes;
  dst->extend_counts.instance_visits += src->extend_counts.instance_visits;
  dst->extend_counts.blas_nodes += src->extend_counts.blas_nodes;
  dst->extend_counts.triangles += src->extend_counts.triangles;
  dst->connect_counts.rays += src->connect_counts.rays;
  dst->connect_counts.tlas_nodes += src->connect_counts.tlas_nodes;
  dst->connect_counts.instance_visits += src->connect_counts.instance_visits;
  dst->connect_counts.blas_nodes += src->connect_counts.blas_nodes;
  dst->connect_counts.triangles += src->connect_counts.triangles;
}

#include "oracle_classic.inc"

/* One bounce of one path: extend (+ connect) + shade.  Returns 1 while the
 * path stays alive. */
static int advance_path(const oracle_scene *s, path *p, const MirtParams *prm, OracleRenderStats *st) {
  tcounts c = {0, 0, 0, 0};
  st->extend_rays++;
  int hit = prm->shade_mode == MIRT_SHADE_CLASSIC ? extend_path_classic(s, p, prm, &c) : extend_path(s, p, prm, &c);
  add_counts(&st->extend_counts, &c, 1);
  if (!hit) return 0;
  if (prm->shade_mode == MIRT_SHADE_CLASSIC) return shade_classic(s, p, prm, st);
  st->shade_events++;
  return shade_uber(s, p);
}

void oracle_render(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                   uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                   uint32_t first_sample, uint32_t spp, uint32_t *sum4, uint32_t *count,
                   OracleRenderStats *stats, int threads) {
  oracle_render_active(s, camera, sources, n_sources, width, height, params, first_sample, spp, NULL, n_sources, sum4,
                       count, NULL, stats, threads);
}

void oracle_render_active(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                          uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                          uint32_t first_sample, uint32_t spp, const uint32_t *active, uint32_t n_active,
                          uint32_t *sum4, uint32_t *count, uint64_t *moments, OracleRenderStats *stats, int threads) {
  (void)n_sources;
  g_moments = moments;
  OracleRenderStats total;
  memset(&total, 0, sizeof(total));
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  const long n_total = (long)n_active * (long)spp;
#pragma omp parallel
  {
    OracleRenderStats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 64)
    for (long i = 0; i < n_total; ++i) {
      const uint32_t source = active ? active[i % n_active] : (uint32_t)(i % n_active);
      const uint32_t k = first_sample + (uint32_t)(i / n_active);
      path p;
      memset(&p, 0, sizeof(p));
      spawn_sample(&p, camera, &sources[source], source, k, width, height, params);
      while (advance_path(s, &p, params, &st)) {
      }
      st.overflowed_sums += accumulate_sample(&p, params, sum4, count);
      st.finished_samples++;
    }
#pragma omp critical
    stats_add(&total, &st);
  }
  if (stats) stats_add(stats, &total);
  g_moments = NULL;
}

/* The convergence test of adaptive sampling, in IEEE doubles operation by
 * operation: with n samples and integer moments S1, S2 of a source,
 * n*S2 - S1^2 <= thr^2 * S1^2 * (n - 1)  (standard error of the mean at most
 * thr x the mean).  Sets bit 0 of flags[i]; returns the sources left unflagged. */
uint32_t oracle_adapt_update(const uint32_t *count, const uint64_t *moments, uint32_t *flags, uint32_t n_sources,
                             float rel_threshold, uint32_t min_samples) {
  const double thr2 = (double)rel_threshold * (double)rel_threshold;
  uint32_t left = 0;
  for (uint32_t i = 0; i < n_sources; ++i) {
    const uint32_t n = count[i];
    if (n >= min_samples && n >= 2u) {
      const double S1 = (double)moments[2 * (size_t)i], S2 = (double)moments[2 * (size_t)i + 1];
      const double S1sq = S1 * S1;
      const double lhs = (double)n * S2 - S1sq;
      const double rhs = thr2 * S1sq * (double)(n - 1u);
      if (lhs <= rhs) flags[i] |= 1u;
    }
    if ((flags[i] & 1u) == 0u) left++;
  }
  return left;
}

static int cmp_u32(const void *a, const void *b) {
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* queuePush / queueRead (shaders/queue.slang:13-36) on one thread */
typedef struct { int count; uint32_t *data; } lifo;
static void queue_push(lifo *q, uint32_t v) { int idx = q->count; q->count += 1; q->data[idx] = v; }
static int queue_read(lifo *q) {
  int idx = q->count; q->count -= 1; idx = idx - 1;
  if (idx < 0) { if (q->count < 0) q->count = 0; return -1; }
  return (int)q->data[idx];
}

/* `limit` samples are handed out at most; the loop ends when the queues are
 * empty or after `max_frames` frames, whichever comes first. */
static void wavefront_run(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                          uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                          uint32_t first_sample, uint64_t limit, uint64_t max_frames, uint32_t slots, uint32_t *sum4,
                          uint32_t *count, OracleRenderStats *stats, uint64_t *frames_out,
                          const uint32_t (*slot_rng)[4]) {
  OracleRenderStats st;
  memset(&st, 0, sizeof(st));
  path *paths = (path *)calloc(slots, sizeof(path));
  /* per-slot streams: the host's seeds (src/pathtracer_state.rs:98-110) */
  if (slot_rng)
    for (uint32_t i = 0; i < slots; ++i) memcpy(paths[i].rng, slot_rng[i], 16);
  uint32_t *ended = (uint32_t *)malloc(4 * (size_t)(slots + 1));
  uint8_t *live = (uint8_t *)calloc(slots, 1);
  lifo term = {0, (uint32_t *)malloc(4 * (size_t)(slots + 1))};
  lifo ext = {0, (uint32_t *)malloc(4 * (size_t)(slots + 1))};
  lifo shade = {0, (uint32_t *)malloc(4 * (size_t)(slots + 1))};
  /* The reference starts with the terminate queue full of zeroed samples,
   * which credits `slots` bogus samples to source 0 (src/queue.rs:14-18,
   * shaders/sample.slang:38); here the first frame only spawns. */
  for (uint32_t i = 0; i < slots; ++i) queue_push(&term, i);
  uint64_t sample_index = 0, frames = 0;
  while ((term.count > 0 || ext.count > 0 || shade.count > 0) && frames < max_frames) {
    /* sampleMain: every thread pops at most one slot.  With per-slot streams
     * the pairing of slots and sample numbers matters; the reference leaves it
     * to the arrival order of its atomics, here the slots that ended take the
     * next numbers in ascending slot order (one of the runs it can make). */
    uint32_t n_ended = 0;
    for (uint32_t th = 0; th < slots; ++th) {
      int idx = queue_read(&term);
      if (idx < 0) break;
      ended[n_ended++] = (uint32_t)idx;
    }
    if (slot_rng) qsort(ended, n_ended, 4, cmp_u32);
    for (uint32_t e = 0; e < n_ended; ++e) {
      const int idx = (int)ended[e];
      if (live[idx]) {
        st.overflowed_sums += accumulate_sample(&paths[idx], params, sum4, count);
        st.finished_samples++;
        live[idx] = 0;
      }
      if (sample_index < limit) {
        uint64_t si = sample_index++;
        uint32_t source = (uint32_t)(si % n_sources), k = first_sample + (uint32_t)(si / n_sources);
        spawn_sample(&paths[idx], camera, &sources[source], source, k, width, height, params);
        live[idx] = 1;
        queue_push(&ext, (uint32_t)idx);
      }
    }
    /* extensionMain */
    for (uint32_t th = 0; th < slots; ++th) {
      int idx = queue_read(&ext);
      if (idx < 0) break;
      tcounts c = {0, 0, 0, 0};
      st.extend_rays++;
      int hit = params->shade_mode == MIRT_SHADE_CLASSIC ? extend_path_classic(s, &paths[idx], params, &c)
                                                        : extend_path(s, &paths[idx], params, &c);
      add_counts(&st.extend_counts, &c, 1);
      if (hit) queue_push(&shade, (uint32_t)idx); else queue_push(&term, (uint32_t)idx);
    }
    /* shadeMain (classic: connect + per-material kernels) */
    for (uint32_t th = 0; th < slots; ++th) {
      int idx = queue_read(&shade);
      if (idx < 0) break;
      int alive;
      if (params->shade_mode == MIRT_SHADE_CLASSIC) alive = shade_classic(s, &paths[idx], params, &st);
      else { st.shade_events++; alive = shade_uber(s, &paths[idx]); }
      if (alive) queue_push(&ext, (uint32_t)idx); else queue_push(&term, (uint32_t)idx);
    }
    frames++;
  }
  if (frames_out) *frames_out = frames;
  if (stats) stats_add(stats, &st);
  free(paths); free(live); free(term.data); free(ext.data); free(shade.data); free(ended);
}

void oracle_render_wavefront(const oracle_scene *s, const MirtCamera *camera,
                             const MirtSampleSource *sources, uint32_t n_sources, uint32_t width,
                             uint32_t height, const MirtParams *params, uint32_t first_sample,
                             uint32_t spp, uint32_t slots, uint32_t *sum4, uint32_t *count,
                             OracleRenderStats *stats, uint64_t *frames_out) {
  wavefront_run(s, camera, sources, n_sources, width, height, params, first_sample, (uint64_t)n_sources * spp,
                ~(uint64_t)0, slots, sum4, count, stats, frames_out, NULL);
}

void oracle_wavefront_per_slot(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                               uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                               const uint32_t (*slot_rng)[4], uint64_t sample_limit, uint64_t n_frames, uint32_t slots,
                               uint32_t *sum4, uint32_t *count, OracleRenderStats *stats) {
  wavefront_run(s, camera, sources, n_sources, width, height, params, 0, sample_limit, n_frames, slots, sum4, count,
                stats, NULL, slot_rng);
}

void oracle_step_wavefront(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                           uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                           uint32_t first_sample, uint64_t n_frames, uint32_t slots, uint32_t *sum4,
                           uint32_t *count, OracleRenderStats *stats) {
  wavefront_run(s, camera, sources, n_sources, width, height, params, first_sample, ~(uint64_t)0, n_frames, slots,
                sum4, count, stats, NULL, NULL);
}

void oracle_bounce_rays(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                        uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                        uint32_t sample_no, uint32_t bounce, MirtRay *out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < (long)n_sources; ++i) {
    OracleRenderStats st;
    memset(&st, 0, sizeof(st));
    path p;
    memset(&p, 0, sizeof(p));
    spawn_sample(&p, camera, &sources[i], (uint32_t)i, sample_no, width, height, params);
    int alive = 1;
    for (uint32_t b = 0; b < bounce && alive; ++b) alive = advance_path(s, &p, params, &st);
    MirtRay r;
    memset(&r, 0, sizeof(r));
    if (alive) {
      r.pos[0] = p.ray_pos.x; r.pos[1] = p.ray_pos.y; r.pos[2] = p.ray_pos.z;
      r.dir[0] = p.ray_dir.x; r.dir[1] = p.ray_dir.y; r.dir[2] = p.ray_dir.z;
      r.last_inst = p.last_inst; r.last_prim = p.last_prim;
    } else {
      r.last_inst = NO_HIT; r.last_prim = NO_HIT;
    }
    out[i] = r;
  }
}

void oracle_connect_rays(const oracle_scene *s, const MirtCamera *camera, const MirtSampleSource *sources,
                         uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                         uint32_t sample_no, uint32_t bounce, MirtRay *out, float *tmax_out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < (long)n_sources; ++i) {
    OracleRenderStats st;
    memset(&st, 0, sizeof(st));
    path p;
    memset(&p, 0, sizeof(p));
    spawn_sample(&p, camera, &sources[i], (uint32_t)i, sample_no, width, height, params);
    int alive = 1, advanced = 0;
    for (uint32_t b = 0; b <= bounce && alive; ++b) {
      p.connect_valid = 0;
      alive = advance_path(s, &p, params, &st);
      advanced = (int)b + 1;
    }
    MirtRay r;
    memset(&r, 0, sizeof(r));
    r.last_inst = NO_HIT; r.last_prim = NO_HIT;
    tmax_out[i] = 0.0f;
    if (advanced == (int)bounce + 1 && p.connect_valid) {
      r = p.connect_ray;
      tmax_out[i] = p.connect_tmax;
    }
    out[i] = r;
  }
}

/* acesToneMap + packRgb (shaders/sample.slang:18-33,52-56, common.slang:177-183).
 * The reference divides by the count BEFORE its own increment (display-only
 * quirk); the mean over the final count is used here. */
static inline float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
void oracle_resolve(const uint32_t *sum4, const uint32_t *count, const MirtSampleSource *sources,
                    uint32_t n_sources, uint32_t width, uint32_t height, const MirtParams *params,
                    uint32_t *rgb) {
  static const float m1[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83770f}};
  static const float m2[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
  for (uint32_t i = 0; i < n_sources; ++i) {
    const uint32_t ox = sources[i].out_pos[0], oy = sources[i].out_pos[1];
    if (ox >= width || oy >= height) continue;
    const float denom = params->fixed_point_scale * (float)count[i];
    float rad[3], v[3], c[3], o[3];
    for (int k = 0; k < 3; ++k) rad[k] = ((float)sum4[4 * (size_t)i + k] / denom) * params->exposure;
    for (int k = 0; k < 3; ++k) v[k] = m1[k][0] * rad[0] + m1[k][1] * rad[1] + m1[k][2] * rad[2];
    for (int k = 0; k < 3; ++k) {
      float a = v[k] * (v[k] + 0.0245786f) - 0.000090537f;
      float b = v[k] * (0.983729f * v[k] + 0.4329510f) + 0.238081f;
      c[k] = a / b;
    }
    for (int k = 0; k < 3; ++k) o[k] = saturate(m2[k][0] * c[0] + m2[k][1] * c[1] + m2[k][2] * c[2]);
    /* packRgb swizzles .bgr first, so red lands in the low byte */
    uint32_t r = (uint32_t)(saturate(o[0]) * 255.0f), g = (uint32_t)(saturate(o[1]) * 255.0f),
             bl = (uint32_t)(saturate(o[2]) * 255.0f);
    if (count[i] == 0) r = g = bl = 0;
    rgb[(size_t)ox + (size_t)oy * width] = (bl << 16) | (g << 8) | r;
  }
}

void oracle_math_selftest(const float *a, const float *b, const float *c, size_t n, float *out6) {
  for (size_t i = 0; i < n; ++i) {
    out6[6 * i + 0] = a[i] + b[i];
    out6[6 * i + 1] = a[i] - b[i];
    out6[6 * i + 2] = a[i] * b[i];
    out6[6 * i + 3] = a[i] / b[i];
    out6[6 * i + 4] = sqrtf(fabsf(a[i]));
    out6[6 * i + 5] = fmaf(a[i], b[i], c[i]);
  }
}
