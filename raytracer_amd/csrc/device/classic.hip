// classic.hip — the per-material shading path: the connect (shadow) kernel
// with light sampling, and one kernel per material kind (Lambertian, Metallic,
// Dielectric, Emissive), each popping its own queue.
//
// Restates shaders/archive/{shadow,light,material,lambertian,metallic,
// dielectric,emissive}.slang on top of the live traversal.  Deliberate
// differences from the archive text, each a repair of something that cannot
// work as written:
//   - the light face is drawn uniformly from the primitive range of the
//     light's BLAS root with its three real vertices (archive/light.slang:64-67
//     adds node.end and broadcasts one scalar to all three vertex ids); the
//     pdf's area is taken from the transformed edges, and (r1,r2) is folded
//     back into the triangle when r1+r2 > 1;
//   - visibility is an any-hit query limited to the light distance that
//     excludes the sampled light triangle, in place of a closest-hit ray that
//     is accepted when it ends within 1.0 of the sample
//     (archive/shadow.slang:100-121); the light's geometric normal replaces
//     the interpolated normal of that closest hit;
//   - the dielectric kernel uses the front_face flag stored by the extension
//     stage (archive/dielectric.slang:66 recomputes it from the already
//     flipped normal, which always yields "front") and multiplies the
//     throughput by the albedo (archive/dielectric.slang:91 assigns it).
#include <float.h>

#include "device_util.h"
#include "sample_stage.h"

namespace mirt {

namespace {

struct SlotIn {
  float4 pa, pb;  // radiance+bounces, throughput+source
  Rng rng;
  v3 rdir;
  v3 hpos, hn;  // hit position, normal flipped to face the ray
  uint32_t inst, ff;
  float tu, tv;  // texture coordinate of the hit (scenes with textures only)
};

// RAW: the hit record still holds what the traversal left (u, v, triangle, instance) and the world-space record of extensionMain
// (shaders/ray_extend.slang:34-50, 204-207) is built here; so is the bounce
// count of archive/extension.slang:94-101 — a hit costs one bounce, and a path
// out of bounces ends without being shaded: the function then returns false
// with only s.pa loaded.
template <bool RAW>
__device__ __forceinline__ bool load_slot(const DScene &sc, const DState &st, uint32_t slot, SlotIn &s) {
  s.pa = path_q(st, slot, 0u);
  if (RAW) {
    const uint32_t bounces = __float_as_uint(s.pa.w) - 1u;
    s.pa.w = __uint_as_float(bounces);
    if (bounces == 0u) return false;
  }
  s.pb = path_q(st, slot, 1u);
  s.rng = load_rng(st, slot);
  v3 rpos;
  uint32_t li, lp;
  load_ray(st.ray, slot, rpos, s.rdir, li, lp);
  s.tu = __uint_as_float(li);  // !RAW: left there by k_extend_finish when the scene has textures
  s.tv = __uint_as_float(lp);
  const float4 *h4 = reinterpret_cast<const float4 *>(st.hit + slot);
  v3 hn;
  if (RAW) {
    const float4 raw = h4[0];
    BestHit best;
    best.u = raw.x;
    best.v = raw.y;
    best.t = 0.0f;
    best.prim = __float_as_uint(raw.z);
    best.inst = __float_as_uint(raw.w);
    finish_hit(sc, s.rdir, best, s.hpos, hn, s.ff);
    s.inst = best.inst;
    if (sc.tri_uv != nullptr) hit_uv(sc, best, s.tu, s.tv);
  } else {
    const float4 ha = h4[0], hb = h4[1];
    const uint32_t inst_ff = __float_as_uint(ha.w);
    s.inst = inst_ff & 0x7FFFFFFFu;
    s.ff = inst_ff >> 31;
    s.hpos = V3(ha.x, ha.y, ha.z);
    hn = V3(hb.x, hb.y, hb.z);
  }
  const float sgn = s.ff != 0u ? 1.0f : -1.0f;
  s.hn = V3(hn.x * sgn, hn.y * sgn, hn.z * sgn);
  return true;
}

__device__ __forceinline__ void store_slot(const DState &st, uint32_t slot, const SlotIn &s, v3 new_pos, v3 new_dir) {
  path_q(st, slot, 0u) = s.pa;
  path_q(st, slot, 1u) = s.pb;
  store_rng(st, slot, s.rng);
  // origins are offset instead of excluding the primitive just hit
  store_ray(st.ray, slot, new_pos, new_dir, kNoPrim, kNoPrim);
}

__device__ __forceinline__ v3 rand_unit_sphere(float a, float b, float c) {
  return normalize3(V3(1.0f - 2.0f * a, 1.0f - 2.0f * b, 1.0f - 2.0f * c));
}

// LambertianData::brdf / pdf (archive/lambertian.slang:15-24)
__device__ __forceinline__ v3 lambert_brdf(v3 albedo, v3 wi, v3 wo, v3 n) {
  if (dot3(n, wi) <= 0.0f || dot3(n, wo) <= 0.0f) return V3(0.0f, 0.0f, 0.0f);
  return V3(albedo.x / 3.1415f, albedo.y / 3.1415f, albedo.z / 3.1415f);
}
__device__ __forceinline__ float lambert_pdf(v3 wi, v3 n) { return fmax_nn(0.0f, dot3(wi, n)) / 3.1415f; }

// ---------------------------------------------------------------- connect
// shadowMain (archive/shadow.slang:63-139) + get_light_sample / tri_sample /
// tri_pdf (archive/light.slang:14-81).  Pops the connect queue, traces the
// shadow ray as an any-hit query, writes the slot's ShadowData and routes the
// slot to the queue of its hit material.
template <bool COUNT>
__global__ MIRT_ENGINE_BOUNDS void k_connect(DScene sc, DState st, DFrameParams fp) {
  const uint32_t *cq = st.qdata[Q_CONNECT];
  const uint32_t count = st.qcount[Q_CONNECT];
  LaneCounts lc;
  unsigned long long traced = 0;

  // Light sampling happens when the slot is fetched.  A slot whose sample lies
  // behind its surface traces nothing: it gets a ray that cannot hit anything
  // (tmax < 0 rejects the TLAS root) and is not counted.
  auto fetch = [&](uint32_t slot, RayIn &r) {
    const float4 *h4 = reinterpret_cast<const float4 *>(st.hit + slot);
    const float4 ha = h4[0], hb = h4[1];
    const uint32_t inst_ff = __float_as_uint(ha.w);
    const float sgn = (inst_ff >> 31) != 0u ? 1.0f : -1.0f;
    const v3 nf = V3(hb.x * sgn, hb.y * sgn, hb.z * sgn);
    const v3 orig = add3(V3(ha.x, ha.y, ha.z), scale3(nf, 1e-3f));
    Rng rng = load_rng(st, slot);
    // light instance and face (two draws), then a point on the face (two draws)
    const float r1 = rng_next(rng);
    const float r2 = rng_next(rng);
    uint32_t li = (uint32_t)__builtin_floorf(r1 * (float)sc.n_lights);
    if (li >= sc.n_lights) li = sc.n_lights - 1u;  // r1 can be exactly 1
    const uint32_t light_inst = sc.lights[li];
    const DInst *inst = &sc.inst[light_inst];
    const uint32_t n_faces = inst->root_count;
    uint32_t fi = (uint32_t)__builtin_floorf(r2 * (float)n_faces);
    if (fi >= n_faces) fi = n_faces - 1u;
    const uint32_t light_prim = inst->root_start + fi;
    const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri) + 3 * (size_t)(inst->tri_base + light_prim);
    const float4 a0 = tri4[0], a1 = tri4[1], a2 = tri4[2];
    const v3 p0 = V3(a0.x, a0.y, a0.z), e1 = V3(a1.x, a1.y, a1.z), e2 = V3(a2.x, a2.y, a2.z);
    float a = rng_next(rng);
    float b = rng_next(rng);
    if (a + b > 1.0f) {
      a = 1.0f - a;
      b = 1.0f - b;
    }
    store_rng(st, slot, rng);
    if (sc.tri_uv != nullptr) {
      // scenes with textures: where on which light triangle the sample lies rides to k_connect_finish
      // in the ray record's origin, which nothing reads between the traversal and the material kernel
      reinterpret_cast<float *>(st.ray + slot)[0] = a;
      reinterpret_cast<float *>(st.ray + slot)[1] = b;
      reinterpret_cast<uint32_t *>(st.ray + slot)[2] = light_prim;
    }
    const v3 sample_pos = xform_point(inst->m, add3(add3(p0, scale3(e1, a)), scale3(e2, b)));
    const v3 cr = cross3(xform_vector(inst->m, e1), xform_vector(inst->m, e2));
    const float area = 0.5f * length3(cr);
    const float tri_pdf = area > 0.0f ? 1.0f / area : 1.0f;
    const float sample_prob = tri_pdf * (1.0f / (float)(sc.n_lights * n_faces));
    const v3 light_n = normalize3(cr);
    const v3 d = sub3(sample_pos, orig);
    const float shadow_t = length3(d);
    const v3 scatter = normalize3(d);
    const bool shoot = dot3(scatter, nf) > 0.0f;
    // what the epilogue needs is parked in the slot's ShadowData
    float4 *sh4 = reinterpret_cast<float4 *>(st.shadow + slot);
    const float prob = sample_prob * dot3(d, d) / __builtin_fabsf(dot3(light_n, neg3(scatter)));
    sh4[0] = make_float4(scatter.x, scatter.y, scatter.z, shoot ? prob : 0.0f);
    sh4[1] = make_float4(__uint_as_float(light_inst), shoot ? 1.0f : 0.0f, 0.0f, 0.0f);
    r.pos = orig;
    r.dir = scatter;
    r.last_inst = light_inst;
    r.last_prim = light_prim;
    r.tmax = shoot ? shadow_t : -1.0f;
    r.payload = slot;
    if (shoot) traced++;
  };
  // the verdict joins what fetch parked; k_connect_finish does the rest
  auto finish = [&](bool fin, uint32_t slot, bool occluded, float, const BestHit &) {
    if (fin) reinterpret_cast<float *>(st.shadow + slot)[7] = occluded ? 1.0f : 0.0f;
  };
  trace_engine<true, COUNT>(sc, WorkSource{&st.qcount[W_CONNECT], count, cq, nullptr, nullptr}, lc, fetch, finish);
  if (COUNT) flush_counts(st.stats, S_CON_TLAS, lc);
  if (COUNT) flush_wave_scan_counts(st.stats, S_CON_TRI_LDS, lc);
  for (int off = 32; off > 0; off >>= 1) traced += __shfl_down(traced, off);
  if (lane_id() == 0u && traced) atomicAdd(&st.stats[S_CONNECT_RAYS], traced);
}

// The epilogue of shadowMain (archive/shadow.slang:122-139), one thread per
// entry of the connect queue: the slot's ShadowData from the parked sample and
// the any-hit verdict, then the slot goes to the queue of its hit material.
__global__ __launch_bounds__(kStreamBlock) void k_connect_finish(DScene sc, DState st, DFrameParams fp) {
  // queues 0..3: the four material queues in QueueId order; 4: terminate
  __shared__ BlockPushN<5> sh;
  const uint32_t par = fp.frame & 1u;
  const uint32_t *cq = st.qdata[Q_CONNECT];
  const uint32_t count = st.qcount[Q_CONNECT];
  uint32_t *const counts[5] = {&st.qcount[Q_LAMBERTIAN], &st.qcount[Q_METALLIC], &st.qcount[Q_DIELECTRIC],
                               &st.qcount[Q_EMISSIVE], &st.qcount[Q_TERM0 + (par ^ 1u)]};
  uint32_t *const datas[5] = {st.qdata[Q_LAMBERTIAN], st.qdata[Q_METALLIC], st.qdata[Q_DIELECTRIC],
                              st.qdata[Q_EMISSIVE], st.qdata[Q_TERM0 + (par ^ 1u)]};
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {  // uniform per block
    const uint32_t i = base + threadIdx.x;
    int route = -1;
    uint32_t slot = 0;
    if (i < count) {
      slot = cq[i];
      float4 *sh4 = reinterpret_cast<float4 *>(st.shadow + slot);
      float4 s0 = sh4[0];
      const float4 parked = sh4[1];  // light instance, shoot flag, light triangle, occluded flag
      const uint32_t light_inst = __float_as_uint(parked.x);
      float4 s1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (parked.y != 0.0f && parked.w == 0.0f) {
        const DInst *linst = &sc.inst[light_inst];
        if (linst->kind == MIRT_MAT_EMISSIVE) {
          const float *e = sc.emissive + 4 * (size_t)linst->kind_data;
          s1 = make_float4(e[0], e[1], e[2], 0.0f);
          const uint32_t tex = sc.tri_uv != nullptr ? sc.mat[linst->material].tex[1] : 0u;
          if (tex != 0u) {
            // an emissive texture on the light: the texel at the sampled point
            BestHit lb;
            lb.t = 0.0f;
            lb.u = reinterpret_cast<const float *>(st.ray + slot)[0];
            lb.v = reinterpret_cast<const float *>(st.ray + slot)[1];
            lb.prim = reinterpret_cast<const uint32_t *>(st.ray + slot)[2];
            lb.inst = light_inst;
            float tu, tv;
            hit_uv(sc, lb, tu, tv);
            const v3 c = tex_sample(sc, tex, tu, tv);
            s1 = make_float4(s1.x * c.x, s1.y * c.y, s1.z * c.z, 0.0f);
          }
        }
      } else {
        s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
      sh4[0] = s0;
      sh4[1] = s1;
      const uint32_t inst_ff = __float_as_uint(reinterpret_cast<const float4 *>(st.hit + slot)[0].w);
      const uint32_t kind = sc.inst[inst_ff & 0x7FFFFFFFu].kind;
      // no material (archive/logic.slang:171-174): the path ends here
      route = kind == MIRT_MAT_NONE ? 4 : (int)kind - (int)MIRT_MAT_LAMBERTIAN;
    }
    block_push_n<5>(sh, counts, datas, route, slot);
  }
}

// One shade event of material kind KIND for the slot whose state is in `s` (process_material and the four
// material files of shaders/archive): radiance, throughput and stream state are updated in `s`, the bounce ray
// comes back in new_pos / new_dir.  Returns whether the path goes on.  `use_shadow`: the connect stage ran and
// left the slot's ShadowData (light sampling).
template <int KIND>
__device__ __forceinline__ bool shade_material(const DScene &sc, const DState &st, uint32_t slot, SlotIn &s, bool use_shadow,
                                               v3 &new_pos, v3 &new_dir) {
  const DInst *inst = &sc.inst[s.inst];
  new_pos = s.hpos;
  new_dir = s.rdir;
  bool alive = false;
  v3 thr = V3(s.pb.x, s.pb.y, s.pb.z), rad = V3(s.pa.x, s.pa.y, s.pa.z);
  // the instance's live material may name a texture: it multiplies this kernel's albedo
  // (the emissive texture for the emissive kernel, the colour texture for the others)
  v3 texel = V3(1.0f, 1.0f, 1.0f);
  if (sc.tri_uv != nullptr) {
    const uint32_t tex = sc.mat[inst->material].tex[KIND == MIRT_MAT_EMISSIVE ? 1 : 0];
    if (tex != 0u) texel = tex_sample(sc, tex, s.tu, s.tv);
  }
  if (KIND == MIRT_MAT_LAMBERTIAN) {
    // process_material<LambertianData> (archive/material.slang:17-139,
    // archive/lambertian.slang:26-35)
    const float *ad = sc.lambertian + 4 * (size_t)inst->kind_data;
    const v3 albedo = V3(ad[0] * texel.x, ad[1] * texel.y, ad[2] * texel.z);
    const v3 nf = s.hn;
    const v3 wo = neg3(s.rdir);
    // scatter draws three values but stores the state after the second
    const float a = rng_next(s.rng);
    const float b = rng_next(s.rng);
    Rng tmp = s.rng;
    const float c = rng_next(tmp);
    const v3 scv = add3(nf, rand_unit_sphere(a, b, c));
    const v3 wi = normalize3(length3(scv) < 1e-6f ? nf : scv);
    const float scatter_prob = lambert_pdf(wi, nf);
    (void)rng_next(s.rng);  // archive/material.slang:91-92
    alive = true;
    if (use_shadow) {
      const float4 *sh4 = reinterpret_cast<const float4 *>(st.shadow + slot);
      const float4 s0 = sh4[0], s1 = sh4[1];
      const float sprob = s0.w;
      if (sprob > 0.0f) {
        const v3 sdir = V3(s0.x, s0.y, s0.z);
        float shadow_weight = 1.0f * sprob;
        shadow_weight /= (1.0f * sprob + 1.0f * lambert_pdf(sdir, nf));
        v3 contrib = thr;
        contrib = mul3(contrib, lambert_brdf(albedo, sdir, wo, nf));
        contrib = scale3(contrib, dot3(sdir, nf));
        contrib = scale3(contrib, shadow_weight);
        const float dl = 1.0f * sprob;
        contrib = V3(contrib.x / dl, contrib.y / dl, contrib.z / dl);
        rad = add3(rad, mul3(V3(s1.x, s1.y, s1.z), contrib));
      }
    }
    if (scatter_prob > 0.0f) {
      float scatter_weight = 1.0f * scatter_prob;
      scatter_weight /= 1.0f * scatter_prob + 1.0f * 0.0f;
      thr = mul3(thr, lambert_brdf(albedo, wi, wo, nf));
      thr = scale3(thr, dot3(wi, nf));
      thr = scale3(thr, scatter_weight);
      const float ds = 1.0f * scatter_prob;
      thr = V3(thr.x / ds, thr.y / ds, thr.z / ds);
    } else {
      thr = scale3(thr, 0.0f);
      alive = false;
    }
    new_pos = add3(s.hpos, scale3(nf, 1e-3f));
    new_dir = wi;
  } else if (KIND == MIRT_MAT_METALLIC) {
    // archive/metallic.slang:65-96
    const float *md = sc.metallic + 8 * (size_t)inst->kind_data;
    const v3 nf = s.hn;
    new_pos = add3(s.hpos, scale3(nf, 0.005f));
    thr = mul3(thr, V3(md[0] * texel.x, md[1] * texel.y, md[2] * texel.z));
    const v3 dir = s.rdir;
    const v3 reflected = sub3(dir, scale3(nf, 2.0f * dot3(dir, nf)));
    const float a = rng_next(s.rng), b = rng_next(s.rng), c = rng_next(s.rng);
    const v3 scv = add3(reflected, scale3(rand_unit_sphere(a, b, c), md[4]));
    new_dir = normalize3(length3(scv) < 1e-2f ? reflected : scv);
    alive = true;
  } else if (KIND == MIRT_MAT_DIELECTRIC) {
    // archive/dielectric.slang:34-95
    const float *dd = sc.dielectric + 8 * (size_t)inst->kind_data;
    const v3 nf = s.hn;
    const float ir = dd[4];
    const float ratio = s.ff != 0u ? (1.0f / ir) : ir;
    const v3 ud = normalize3(s.rdir);
    const float cos_theta = fmin_nn(-dot3(ud, nf), 1.0f);
    const float sin_theta = __builtin_sqrtf(fmax_nn(0.0f, 1.0f - pow2f(cos_theta)));
    const bool cannot_refract = ratio * sin_theta > 1.0f;
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 *= r0;
    const float reflectance = r0 + (1.0f - r0) * pow5f(1.0f - cos_theta);
    const float rnd = rng_next(s.rng);
    v3 scv;
    if (cannot_refract || reflectance > rnd) {
      scv = sub3(ud, scale3(nf, 2.0f * dot3(ud, nf)));
      new_pos = add3(s.hpos, scale3(nf, 1e-4f));
    } else {
      const float ct = fmin_nn(-dot3(ud, nf), 1.0f);
      const v3 perp = scale3(add3(ud, scale3(nf, ct)), ratio);
      const float k = 1.0f - dot3(perp, perp);
      const v3 par_v = scale3(nf, -__builtin_sqrtf(fmax_nn(k, 0.0f)));
      scv = add3(perp, par_v);
      new_pos = sub3(s.hpos, scale3(nf, 1e-4f));
    }
    thr = mul3(thr, V3(dd[0] * texel.x, dd[1] * texel.y, dd[2] * texel.z));
    new_dir = normalize3(scv);
    alive = true;
  } else {
    // archive/emissive.slang:42-59
    const float *ed = sc.emissive + 4 * (size_t)inst->kind_data;
    rad = add3(rad, mul3(V3(ed[0] * texel.x, ed[1] * texel.y, ed[2] * texel.z), thr));
    alive = false;
  }
  s.pa.x = rad.x; s.pa.y = rad.y; s.pa.z = rad.z;
  s.pb.x = thr.x; s.pb.y = thr.y; s.pb.z = thr.z;
  return alive;
}

// ------------------------------------------------------- material kernels
// Common frame of the four kernels: pop `queue`, shade, push to the next
// frame's extension queue (alive) or terminate queue.
#ifndef MIRT_MATERIAL_WAVES
#define MIRT_MATERIAL_WAVES 4
#endif
template <int KIND>
__global__ __launch_bounds__(kStreamBlock, MIRT_MATERIAL_WAVES) void k_material(DScene sc, DState st, DFrameParams fp) {
  __shared__ BlockPush2 sh;
  const uint32_t par = fp.frame & 1u;
  const int q = Q_LAMBERTIAN - 1 + KIND;
  const uint32_t *mq = st.qdata[q];
  const uint32_t count = st.qcount[q];
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool active = i < count;
    uint32_t slot = 0;
    bool alive = false;
    if (active) {
      slot = mq[i];
      SlotIn s;
      (void)load_slot<false>(sc, st, slot, s);
      v3 new_pos, new_dir;
      alive = shade_material<KIND>(sc, st, slot, s, true, new_pos, new_dir);  // the connect kernel ran and left ShadowData
      store_slot(st, slot, s, new_pos, new_dir);
    }
    block_push2(sh, &st.qcount[Q_EXT0 + (par ^ 1u)], st.qdata[Q_EXT0 + (par ^ 1u)], active && alive,
                &st.qcount[Q_TERM0 + (par ^ 1u)], st.qdata[Q_TERM0 + (par ^ 1u)], active && !alive, slot);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&st.stats[S_SHADE_EVENTS], (unsigned long long)count);
}

// Per-material shading without light sampling, in ONE pass over the extension queue: extensionMain's epilogue
// (sky on a miss, the bounce count of archive/extension.slang:94-101) and the hit material's shade event, with
// the traversal's result and the path record in registers between the two — no routing pass, no material
// queues, no hit-record round trip.  A wave whose lanes hit different kinds of material runs their branches one
// after the other; each is a few dozen instructions against the dozen scattered memory requests a slot costs.
// The four k_material kernels stay for light sampling, where the connect stage sits between the two halves.
//
// RESPAWN: a path that ends here — sky, out of bounces, no material, absorbed by its material — is
// accumulated and its slot takes the render call's next sample at once (as in k_finish_shade_uber<true>).
template <bool RESPAWN>
__global__ __launch_bounds__(kStreamBlock, MIRT_MATERIAL_WAVES) void k_finish_shade_classic(DScene sc, DState st, DFrameParams fp) {
  __shared__ BlockPush2 sh;
  __shared__ BlockReserve shr;
  const uint32_t par = fp.frame & 1u;
  const uint32_t *ext = st.qdata[Q_EXT0 + par];
  const uint32_t count = st.qcount[Q_EXT0 + par];
  CamBasis cb;
  if (RESPAWN) cb = cam_basis(fp.cam);
  unsigned long long shaded = 0, finished = 0, spawned = 0;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {  // uniform per block
    const uint32_t i = base + threadIdx.x;
    const bool active = i < count;
    uint32_t slot = 0;
    bool alive = false;
    float4 rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // of a path that ends here
    uint32_t source = kNoSample;
    if (active) {
      slot = ext[i];
      const uint32_t inst = __float_as_uint(reinterpret_cast<const float4 *>(st.hit + slot)[0].w);
      if (inst == kMissInst) {
        // rad += throughput * sky
        float4 pa = path_q(st, slot, 0u);
        const float4 pb = path_q(st, slot, 1u);
        const float sky = fp.p.sky_radiance;
        pa.x = pa.x + pb.x * sky;
        pa.y = pa.y + pb.y * sky;
        pa.z = pa.z + pb.z * sky;
        rad = pa;
        source = __float_as_uint(pb.w);
        if (!RESPAWN) path_q(st, slot, 0u) = pa;
      } else {
        const uint32_t kind = sc.inst[inst].kind;
        SlotIn s;
        if (kind == MIRT_MAT_NONE) {
          // no material (archive/logic.slang:171-174): the path ends as it is
          rad = path_q(st, slot, 0u);
          source = path_w(st, slot, 7u);
        } else if (!load_slot<true>(sc, st, slot, s)) {
          // out of bounces: the path ends unshaded
          rad = s.pa;
          source = path_w(st, slot, 7u);
          if (!RESPAWN) path_q(st, slot, 0u) = s.pa;
        } else {
          v3 new_pos, new_dir;
          if (kind == MIRT_MAT_LAMBERTIAN) alive = shade_material<MIRT_MAT_LAMBERTIAN>(sc, st, slot, s, false, new_pos, new_dir);
          else if (kind == MIRT_MAT_METALLIC) alive = shade_material<MIRT_MAT_METALLIC>(sc, st, slot, s, false, new_pos, new_dir);
          else if (kind == MIRT_MAT_DIELECTRIC) alive = shade_material<MIRT_MAT_DIELECTRIC>(sc, st, slot, s, false, new_pos, new_dir);
          else alive = shade_material<MIRT_MAT_EMISSIVE>(sc, st, slot, s, false, new_pos, new_dir);
          shaded++;
          rad = s.pa;
          source = __float_as_uint(s.pb.w);
          if (alive || !RESPAWN) store_slot(st, slot, s, new_pos, new_dir);
        }
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
        spawn_sample(st, fp, cb, slot, si);
        spawned++;
        alive = true;
      } else if (done) {
        path_w(st, slot, 7u) = kNoSample;  // the slot retires
      }
      block_push2(sh, &st.qcount[Q_EXT0 + (par ^ 1u)], st.qdata[Q_EXT0 + (par ^ 1u)], active && alive, nullptr, nullptr,
                  false, slot);
    } else {
      block_push2(sh, &st.qcount[Q_EXT0 + (par ^ 1u)], st.qdata[Q_EXT0 + (par ^ 1u)], active && alive,
                  &st.qcount[Q_TERM0 + (par ^ 1u)], st.qdata[Q_TERM0 + (par ^ 1u)], active && !alive, slot);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    shaded += __shfl_down(shaded, off);
    if (RESPAWN) {
      finished += __shfl_down(finished, off);
      spawned += __shfl_down(spawned, off);
    }
  }
  if (lane_id() == 0u) {
    if (shaded) atomicAdd(&st.stats[S_SHADE_EVENTS], shaded);
    if (RESPAWN && finished) atomicAdd(&st.stats[S_FINISHED], finished);
    if (RESPAWN && spawned) atomicAdd(&st.stats[S_SPAWNED], spawned);
  }
}

}  // namespace

void launch_finish_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  if (fp.respawn) hipLaunchKernelGGL((k_finish_shade_classic<true>), stream_grid(cfg, k_finish_shade_classic<true>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_finish_shade_classic<false>), stream_grid(cfg, k_finish_shade_classic<false>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
}

void launch_connect(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  const dim3 grid(trace_grid_blocks(cfg.num_cus));
  if (fp.counting) hipLaunchKernelGGL((k_connect<true>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_connect<false>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL(k_connect_finish, stream_grid(cfg, k_connect_finish), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
}

void launch_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  // the four per-material kernels, one queue each (archive frame order:
  // src/app.rs:402-410 dispatches every material phase each frame)
  const dim3 block(kStreamBlock);
  hipLaunchKernelGGL((k_material<MIRT_MAT_LAMBERTIAN>), stream_grid(cfg, k_material<MIRT_MAT_LAMBERTIAN>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_METALLIC>), stream_grid(cfg, k_material<MIRT_MAT_METALLIC>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_DIELECTRIC>), stream_grid(cfg, k_material<MIRT_MAT_DIELECTRIC>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_EMISSIVE>), stream_grid(cfg, k_material<MIRT_MAT_EMISSIVE>), block, 0, cfg.stream, sc, st, fp);
}

}  // namespace mirt
