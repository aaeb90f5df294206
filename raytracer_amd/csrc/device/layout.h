// layout.h — how the scene and the path state are laid out in HBM.
//
// The caller hands over the reference's buffers (64-B nodes, 16-B index rows
// pointing into 48-B vertices, 48-B Euler transforms).  Traversal is a chain
// of dependent loads, so the upload re-packs them so that every step is ONE
// aligned fetch and nothing is recomputed per ray:
//
//   DNode (32 B)   box + two words.  Node i of a tree sits at byte offset
//                  32*(base+i) with base odd, so the two children of a node —
//                  always pushed as the pair (l, l+1) with l odd by the
//                  builder — share one 64-B line.
//   DTri (48 B)    p0, e1 = p1-p0, e2 = p2-p0 of triangle p, indexed by the
//                  BLAS-local triangle id the leaves refer to: one 48-B fetch
//                  per test instead of a 16-B index row plus three 48-B
//                  vertices.  e1/e2 are the same f32 subtractions the kernel
//                  would do, so results are unchanged.
//   scan blocks    a lane-interleaved second copy of the triangles of big leaves
//                  (see kScanLeafMin below).
//   DTriN (48 B)   the three vertex normals, fetched once for the winning
//                  triangle only.
//   DInst (128 B)  world->object and object->world 3x4 matrices (the
//                  reference rebuilds both with six sin/cos and four 4x4
//                  products for every ray x instance visit), the node / triangle
//                  bases and the material of an instance.
//
// Topology, box values, visit order and triangle ids are the builder's,
// untouched: equal-t ties and the set of visited leaves stay identical.
#pragma once

#include <stdint.h>

namespace mirt {

// word b: bits 0..23 skip link (next node when this subtree is done),
//         bits 24..31 leaf size: 0 = inner, 1..254 = leaf with that many
//         primitives, 255 = leaf whose size is stored out of line
//         (BLAS: bits of DTri[start].p0w; TLAS: tlas_big[start]).
// word a: inner -> left child; leaf -> first primitive (BLAS-local triangle id
//         or TLAS slot).
struct DNode {
  float lbx, lby, lbz;
  uint32_t a;
  float ubx, uby, ubz;
  uint32_t b;
};
static_assert(sizeof(DNode) == 32, "DNode must be 32 bytes");

constexpr uint32_t kSkipMask = 0x00FFFFFFu;
constexpr uint32_t kMaxNodesPerTree = 1u << 24;
constexpr uint32_t kBigLeaf = 255u;

struct DTri {
  float p0x, p0y, p0z;
  uint32_t p0w;  // out-of-line leaf size for the first triangle of a big leaf
  float e1x, e1y, e1z;
  uint32_t e1w;  // first triangle of a leaf the whole wave scans: byte offset of its scan blocks
  float e2x, e2y, e2z;
  uint32_t e2w;  // first triangle of a leaf held in the LDS leaf cache: byte offset of its blocks / index rows in the image + 1
};

// Leaves with kScanLeafMin or more triangles are tested by a whole wave, 64
// triangles per step.  They keep a second copy of their triangles in blocks of
// 64, each block three lane-interleaved planes (16 + 16 + 4 bytes per lane), so
// that a step is three contiguous loads instead of three loads of stride 48.
#ifndef MIRT_SCAN_LEAF_MIN
#define MIRT_SCAN_LEAF_MIN 24u
#endif
constexpr uint32_t kScanLeafMin = MIRT_SCAN_LEAF_MIN;
constexpr uint32_t kScanBlockBytes = 2304u;

// The largest of those leaves are also kept in LDS by the traversal kernels
// (their blocks would otherwise be streamed from the L2 by every ray that
// reaches them): one image of at most kLeafCacheBytes, copied into LDS when a
// block starts.  Two forms of image:
//   direct (kLeafCacheDirect)  the leaves' scan blocks as they are in memory, 36 bytes per
//            triangle: a lane reads its triangle with two 16-byte and one 4-byte LDS read
//            and tests it at once;
//   indexed  the unique vertices of the cached leaves (float4 each, w unused), then per
//            leaf one uint2 per triangle holding three 16-bit vertex numbers (x = v0 |
//            v1 << 16, y = v2), about 16 bytes per triangle; the kernel rebuilds
//            e1 = p1 - p0 and e2 = p2 - p0 with the same f32 subtractions the upload
//            does for DTri.
// One engine block runs per CU and the CU has 160 KB of LDS, so the direct form
// gets 128 KB (about 3 600 triangles).
#ifndef MIRT_LEAF_CACHE_DIRECT
#define MIRT_LEAF_CACHE_DIRECT 1
#endif
constexpr bool kLeafCacheDirect = MIRT_LEAF_CACHE_DIRECT != 0;
#ifndef MIRT_LEAF_CACHE_KB
#if MIRT_LEAF_CACHE_DIRECT
#define MIRT_LEAF_CACHE_KB 128u
#else
#define MIRT_LEAF_CACHE_KB 64u
#endif
#endif
constexpr uint32_t kLeafCacheBytes = MIRT_LEAF_CACHE_KB * 1024u;
static_assert(sizeof(DTri) == 48, "DTri must be 48 bytes");

struct DTriN {
  float n0[4], n1[4], n2[4];
};
static_assert(sizeof(DTriN) == 48, "DTriN must be 48 bytes");

struct DInst {
  float mi[12];        // world -> object, rows of a 3x4
  float m[12];         // object -> world
  uint32_t node_base;  // index of this instance's BLAS root in the node array
  uint32_t tri_base;   // index of its triangle 0 in the DTri / DTriN arrays
  uint32_t material;   // index into the material table
  uint32_t kind;       // per-material kernel (MIRT_MAT_*), 0 when unused
  uint32_t kind_data;  // row in that kernel's table
  uint32_t root_start, root_count;  // primitive range of the BLAS root (light sampling)
  uint32_t pad;
};
static_assert(sizeof(DInst) == 128, "DInst must be 128 bytes");

// Per-slot path state.  Records are 16-B multiples so a slot is read with
// whole dwordx4 loads; slots are gathered through the queues' index lists.
struct DRay {  // same bytes as MirtRay (the API's ray lists); the path state's own rays order the eight words
               // differently, see load_pray in device_util.h
  float px, py, pz, dx;
  float dy, dz;
  uint32_t last_inst, last_prim;
};
static_assert(sizeof(DRay) == 32, "DRay must be 32 bytes");

struct DHit {  // world-space hit of the connect stage (per-material shading with light sampling); not allocated otherwise
  float px, py, pz;
  uint32_t inst_ff;  // instance id | front_face << 31
  float nx, ny, nz;
  uint32_t prim;
};
static_assert(sizeof(DHit) == 32, "DHit must be 32 bytes");

// DPath records as three planes of float4 (see path_q in device_util.h) instead of 48-byte records.
// Measured and not used: the queues list slots in the order paths survive, not in slot order (a render call
// hands every slot two or three samples and then retires it, so most frames compact), and a scattered slot
// then costs three cache lines instead of one or two: the uber shade kernel 2.29 -> 2.40 ms per launch, the
// per-material kernels of config 2 1.65 -> 2.25 ms per frame.  With 256 samples per pixel and render call
// (eight samples per slot, queues mostly in slot order) it is the other way round, but barely: the uber kernel
// 4.14 -> 4.08 ms per launch, 1359 -> 1363 Mrays/s.
#ifndef MIRT_PATH_PLANES
#define MIRT_PATH_PLANES 0
#endif
constexpr bool kPathPlanes = MIRT_PATH_PLANES != 0;

struct DPath {
  float rx, ry, rz;   // radiance
  uint32_t bounces;
  float tx, ty, tz;   // throughput
  uint32_t source;    // sample source index; 0xFFFFFFFF = slot holds no sample
  uint32_t rng[4];
};
static_assert(sizeof(DPath) == 48, "DPath must be 48 bytes");

// The connect stage's record of a slot.  First half: archive ShadowData's direction to the light sample and
// its probability (0: nothing shot, or the shadow ray was stopped).  Second half, as quarters of four words:
// between the extend epilogue and the material, the light instance, the distance to the sample (negative:
// nothing shot), the light triangle and the sample's first barycentric; behind the routing pass of
// MirtParams.stage_kernels, ShadowData's radiance.
struct DShadow {
  float dx, dy, dz, prob;
  float rx, ry, rz, pad;
};
static_assert(sizeof(DShadow) == 32, "DShadow must be 32 bytes");

struct DMaterial {  // the fields of MirtMaterial the kernels read
  float colour[4];
  float emissive[4];
  float metallic, roughness, ior, transmission;
  uint32_t tex[4];  // colour texture, emissive texture (0 = none, t = textures[t - 1]), unused, unused
};
static_assert(sizeof(DMaterial) == 64, "DMaterial must be 64 bytes");

// Everything a traversal needs, passed by value to the kernels.
struct DScene {
  const unsigned char *geo;   // one allocation: the node array, then (at tri_off) the DTri array
  uint32_t geo_bytes, tri_off;
  uint32_t tri36_off;         // the triangles again, 36 bytes each (p0, e1, e2 and nothing else), for the lanes' own fetches
  const DNode *nodes;         // = geo; every tree: node i of a tree at nodes[base + i]
  uint32_t tlas_base;         // base of the TLAS
  const uint32_t *tlas_big;   // out-of-line TLAS leaf sizes, indexed by the leaf's first slot
  const uint32_t *tlas_to_inst;
  const DInst *inst;
  const DTri *tri;            // = geo + tri_off
  const DTriN *trin;
  const uint4 *leaf_cache;    // image of the LDS leaf cache (see kLeafCacheBytes)
  uint32_t leaf_cache_bytes;  // multiple of 16, 0 when no leaf is cached
  const DMaterial *mat;
  const uint32_t *lights;     // instance ids
  uint32_t n_lights;          // 0 when the scene has none
  const float *lambertian;    // 4 floats per row (albedo)
  const float *metallic;      // 8 floats per row (albedo, fuzz, pad)
  const float *dielectric;    // 8 floats per row (albedo, ir, pad)
  const float *emissive;      // 4 floats per row (albedo)
  // textures (all null / 0 when the scene has none)
  const float4 *tri_uv;       // two float4 per triangle: (u0, v0, u1, v1), (u2, v2, -, -)
  const uint4 *textures;      // MirtTexture rows: width, height, first texel, filter
  const uint32_t *texels;     // RGBA8
  uint32_t n_textures;
};

struct DCounts {
  unsigned long long rays, tlas_nodes, instance_visits, blas_nodes, triangles;
};

}  // namespace mirt
