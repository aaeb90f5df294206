// classic.hip — the per-material shading path: the extend epilogue with light
// sampling, the connect (shadow) kernel, and the material kinds (Lambertian,
// Metallic, Dielectric, Emissive) — as one kernel behind the connect stage, or,
// with MirtParams.stage_kernels, as one kernel per material popping its own
// queue, the way the archive frame dispatches them.
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
//
// What a slot's records hold between the stages of a frame with light sampling
// (DRay, DHit, DShadow of layout.h, as quarters of 16 bytes):
//   after k_extend            ray[0] = u, v, triangle, instance (or kMissInst); ray[1] = incoming direction xyz, last triangle
//   after k_extend_finish     ray[0] = shadow-ray origin xyz, texture coordinate u of the hit (scenes with textures)
//                             ray[1] = incoming direction xyz, texture coordinate v (else the last triangle, no longer needed)
//                             hit[0] = world position xyz, instance | front_face << 31
//                             hit[1] = world normal xyz, b of the light sample
//                             shadow[0] = direction to the light sample xyz, its probability (0: none shot)
//                             shadow[1] = light instance, distance to the sample (negative: none shot),
//                                         light triangle, a of the light sample
//   after k_connect           shadow[0].w = 0 where the shadow ray was stopped
// so the walk's refill — single lanes, with the rest of the wave waiting — reads three quarters and
// computes nothing, and everything that is arithmetic runs in the streaming kernels.
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
  float hit_w;   // the last word of the world-space hit record (light sampling: b of the light sample)
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
  const float4 *r4 = reinterpret_cast<const float4 *>(st.ray + slot);
  const float4 ra = r4[0], rb = r4[1];
  s.rdir = V3(rb.x, rb.y, rb.z);
  s.tu = ra.w;  // !RAW: left in the ray's two exclusion words by k_extend_finish when the scene has textures
  s.tv = rb.w;
  v3 hn;
  if (RAW) {
    const float4 raw = ra;  // the walk's result (load_pray in device_util.h)
    BestHit best;
    best.u = raw.x;
    best.v = raw.y;
    best.t = 0.0f;
    best.prim = __float_as_uint(raw.z);
    best.inst = __float_as_uint(raw.w);
    finish_hit(sc, s.rdir, best, s.hpos, hn, s.ff);
    s.inst = best.inst;
    s.hit_w = 0.0f;
    if (sc.tri_uv != nullptr) hit_uv(sc, best, s.tu, s.tv);
  } else {
    const float4 *h4 = reinterpret_cast<const float4 *>(st.hit + slot);
    const float4 ha = h4[0], hb = h4[1];
    const uint32_t inst_ff = __float_as_uint(ha.w);
    s.inst = inst_ff & 0x7FFFFFFFu;
    s.ff = inst_ff >> 31;
    s.hpos = V3(ha.x, ha.y, ha.z);
    hn = V3(hb.x, hb.y, hb.z);
    s.hit_w = hb.w;
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
  store_pray(st.ray, slot, new_pos, new_dir, kNoPrim, kNoPrim);
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

// ------------------------------------------------- extend epilogue + light sample
// get_light_sample / tri_sample / tri_pdf (archive/light.slang:14-81) and the shadow ray of shadowMain
// (archive/shadow.slang:63-121) for a hit at `hpos` with the normal `nf` facing the incoming ray: light
// instance and face (two draws), a point on the face (two draws).
struct LightSample {
  v3 orig, scatter;   // the shadow ray
  float dist;         // to the sample; negative when the sample lies behind the surface (nothing is shot)
  float prob;         // solid-angle probability of the sample, 0 when nothing is shot
  uint32_t inst, prim;  // the light and its sampled triangle (the shadow ray must not stop at it)
  float a, b;         // where on the triangle the sample lies
};
__device__ __forceinline__ LightSample sample_light(const DScene &sc, Rng &rng, v3 hpos, v3 nf) {
  LightSample ls;
  ls.orig = add3(hpos, scale3(nf, 1e-3f));
  const float r1 = rng_next(rng);
  const float r2 = rng_next(rng);
  uint32_t li = (uint32_t)__builtin_floorf(r1 * (float)sc.n_lights);
  if (li >= sc.n_lights) li = sc.n_lights - 1u;  // r1 can be exactly 1
  ls.inst = sc.lights[li];
  const DInst *inst = &sc.inst[ls.inst];
  const uint32_t n_faces = inst->root_count;
  uint32_t fi = (uint32_t)__builtin_floorf(r2 * (float)n_faces);
  if (fi >= n_faces) fi = n_faces - 1u;
  ls.prim = inst->root_start + fi;
  const float4 *tri4 = reinterpret_cast<const float4 *>(sc.tri) + 3 * (size_t)(inst->tri_base + ls.prim);
  const float4 a0 = tri4[0], a1 = tri4[1], a2 = tri4[2];
  const v3 p0 = V3(a0.x, a0.y, a0.z), e1 = V3(a1.x, a1.y, a1.z), e2 = V3(a2.x, a2.y, a2.z);
  float a = rng_next(rng);
  float b = rng_next(rng);
  if (a + b > 1.0f) {
    a = 1.0f - a;
    b = 1.0f - b;
  }
  ls.a = a;
  ls.b = b;
  const v3 sample_pos = xform_point(inst->m, add3(add3(p0, scale3(e1, a)), scale3(e2, b)));
  const v3 cr = cross3(xform_vector(inst->m, e1), xform_vector(inst->m, e2));
  const float area = 0.5f * length3(cr);
  const float tri_pdf = area > 0.0f ? 1.0f / area : 1.0f;
  const float sample_prob = tri_pdf * (1.0f / (float)(sc.n_lights * n_faces));
  const v3 light_n = normalize3(cr);
  const v3 d = sub3(sample_pos, ls.orig);
  const float shadow_t = length3(d);
  ls.scatter = normalize3(d);
  const bool shoot = dot3(ls.scatter, nf) > 0.0f;
  const float prob = sample_prob * dot3(d, d) / __builtin_fabsf(dot3(light_n, neg3(ls.scatter)));
  ls.prob = shoot ? prob : 0.0f;
  ls.dist = shoot ? shadow_t : -1.0f;
  return ls;
}

// The extend epilogue of per-material shading with light sampling (shaders/ray_extend.slang:34-50 with the
// bounce count and routing of archive/extension.slang:94-101), one thread per entry of the extension queue:
// a miss adds the sky and ends the path; a hit gets its world-space record, loses one bounce and — with
// bounces left — draws its light sample here, where every lane of a wave has an entry, and goes to the
// connect queue with the shadow ray ready in its records (the table at the top of this file).
// (Uber shading does its epilogue inside k_finish_shade_uber, per-material shading without light sampling
// inside k_finish_shade_classic.)
//
// RESPAWN: a path that ends here (sky, out of bounces) is accumulated and its slot takes the next sample at
// once, as in k_finish_shade_uber<true>.
template <bool RESPAWN>
__global__ __launch_bounds__(kStreamBlock) void k_extend_finish(DScene sc, DState st, DFrameParams fp) {
  const uint32_t par = fp.frame & 1u;
  const uint32_t *ext = st.qdata[Q_EXT0 + par];
  const uint32_t count = st.qcount[Q_EXT0 + par];
  // a path that ends goes to the terminate queue (RESPAWN: its slot, respawned, to the next frame's extension
  // queue); a hit goes to the connect queue
  const int q0 = RESPAWN ? Q_EXT0 + (int)(par ^ 1u) : Q_TERM0 + (int)(par ^ 1u);
  __shared__ BlockPush2 sh;
  __shared__ BlockReserve shr;
  unsigned long long finished = 0, spawned = 0;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {  // uniform per block
    const uint32_t i = base + threadIdx.x;
    const bool fin = i < count;
    uint32_t slot = 0;
    bool hit = false;
    float4 end_rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t end_src = kNoSample;
    bool have_end = false;
    if (fin) {
      slot = ext[i];
      float4 *h4 = reinterpret_cast<float4 *>(st.hit + slot);
      float4 *r4 = reinterpret_cast<float4 *>(st.ray + slot);
      const float4 raw = r4[0];  // the walk's result
      BestHit best;
      best.u = raw.x;
      best.v = raw.y;
      best.t = 0.0f;
      best.prim = __float_as_uint(raw.z);
      best.inst = __float_as_uint(raw.w);
      if (best.inst == kMissInst) {
        // rad += throughput * sky
        float4 a = path_q(st, slot, 0u);
        const float4 b = path_q(st, slot, 1u);
        const float sky = fp.p.sky_radiance;
        a.x = a.x + b.x * sky;
        a.y = a.y + b.y * sky;
        a.z = a.z + b.z * sky;
        if (RESPAWN) {
          end_rad = a;
          end_src = __float_as_uint(b.w);
          have_end = true;
        } else {
          path_q(st, slot, 0u) = a;
        }
      } else {
        hit = true;
        const float4 rb = r4[1];
        v3 hp, hn;
        uint32_t ff;
        finish_hit(sc, V3(rb.x, rb.y, rb.z), best, hp, hn, ff);
        // archive/extension.slang:94-101: a hit costs one bounce; out of
        // bounces means terminate without shading
        uint32_t &pb = path_w(st, slot, 3u);
        const uint32_t bounces = pb - 1u;
        pb = bounces;
        if (bounces == 0u) {
          hit = false;
        } else {
          const float sgn = ff != 0u ? 1.0f : -1.0f;
          Rng rng = load_rng(st, slot);
          const LightSample ls = sample_light(sc, rng, hp, V3(hn.x * sgn, hn.y * sgn, hn.z * sgn));
          store_rng(st, slot, rng);
          float4 *sh4 = reinterpret_cast<float4 *>(st.shadow + slot);
          sh4[0] = make_float4(ls.scatter.x, ls.scatter.y, ls.scatter.z, ls.prob);
          sh4[1] = make_float4(__uint_as_float(ls.inst), ls.dist, __uint_as_float(ls.prim), ls.a);
          h4[0] = make_float4(hp.x, hp.y, hp.z, __uint_as_float(best.inst | (ff << 31)));
          h4[1] = make_float4(hn.x, hn.y, hn.z, ls.b);
          // the material's texture coordinate rides in the ray's two exclusion words,
          // which nothing reads between this kernel and the material's own store
          float tu = 0.0f, tv = 0.0f;
          if (sc.tri_uv != nullptr) hit_uv(sc, best, tu, tv);
          r4[0] = make_float4(ls.orig.x, ls.orig.y, ls.orig.z, tu);
          if (sc.tri_uv != nullptr) r4[1] = make_float4(rb.x, rb.y, rb.z, tv);
        }
      }
      if (RESPAWN && !hit && !have_end) {
        end_rad = path_q(st, slot, 0u);
        end_src = path_w(st, slot, 7u);
      }
    }
    int r = !fin ? -1 : (!hit ? 0 : 1);
    if (RESPAWN) {
      const bool done = fin && !hit;
      if (done) {
        accumulate_sample(st, fp, end_src, end_rad);
        finished++;
      }
      const unsigned long long si = block_reserve(shr, st.sample_index, done);
      if (done && si < fp.sample_limit) {
        spawn_sample(st, fp, cam_basis(fp.cam), slot, si);  // the basis is a few products, cheaper than registers held for it
        spawned++;
      } else if (done) {
        path_w(st, slot, 7u) = kNoSample;  // the slot retires
        r = -1;
      }
    }
    block_push2(sh, &st.qcount[q0], st.qdata[q0], r == 0, &st.qcount[Q_CONNECT], st.qdata[Q_CONNECT], r > 0, slot);
  }
  if (RESPAWN) {
    for (int off = 32; off > 0; off >>= 1) {
      finished += __shfl_down(finished, off);
      spawned += __shfl_down(spawned, off);
    }
    if (lane_id() == 0u) {
      if (finished) atomicAdd(&st.stats[S_FINISHED], finished);
      if (spawned) atomicAdd(&st.stats[S_SPAWNED], spawned);
    }
  }
}

// ---------------------------------------------------------------- connect
// The visibility query of shadowMain (archive/shadow.slang:100-121): pops the connect queue and traces the
// shadow ray k_extend_finish left in the slot's records as an any-hit query.  A slot whose light sample
// lies behind its surface traces nothing: it gets a ray that cannot hit anything (tmax < 0 rejects the TLAS
// root) and is not counted.  The verdict: the sample's probability is cleared where the ray was stopped.
template <bool COUNT>
__global__ MIRT_ENGINE_BOUNDS void k_connect(DScene sc, DState st, DFrameParams fp) {
  const uint32_t *cq = st.qdata[Q_CONNECT];
  const uint32_t count = st.qcount[Q_CONNECT];
  LaneCounts lc;
  unsigned long long traced = 0;
  auto fetch = [&](uint32_t slot, RayIn &r) {
    const float4 o = reinterpret_cast<const float4 *>(st.ray + slot)[0];
    const float4 *sh4 = reinterpret_cast<const float4 *>(st.shadow + slot);
    const float4 s0 = sh4[0], s1 = sh4[1];
    r.pos = V3(o.x, o.y, o.z);
    r.dir = V3(s0.x, s0.y, s0.z);
    r.last_inst = __float_as_uint(s1.x);
    r.last_prim = __float_as_uint(s1.z);
    r.tmax = s1.y;
    r.payload = slot;
    if (!(s1.y < 0.0f)) traced++;
  };
  auto finish = [&](bool fin, uint32_t slot, bool occluded, float, const BestHit &) {
    if (fin && occluded) reinterpret_cast<float *>(st.shadow + slot)[3] = 0.0f;
  };
  trace_engine<true, COUNT>(sc, WorkSource{&st.qcount[W_CONNECT], count, cq, nullptr, nullptr}, lc, fetch, finish);
  if (COUNT) flush_counts(st.stats, S_CON_TLAS, lc);
  if (COUNT) flush_wave_scan_counts(st.stats, S_CON_TRI_LDS, lc);
  for (int off = 32; off > 0; off >>= 1) traced += __shfl_down(traced, off);
  if (lane_id() == 0u && traced) atomicAdd(&st.stats[S_CONNECT_RAYS], traced);
}

// The radiance of the light sample of `slot` (the epilogue of shadowMain, archive/shadow.slang:122-139):
// the light's emission, times its emissive texture at the sampled point when it has one.  `s1` is the
// slot's shadow[1] quarter, `b` rides in hit[1].w.
__device__ __forceinline__ v3 light_radiance(const DScene &sc, float4 s1, float b) {
  const uint32_t light_inst = __float_as_uint(s1.x);
  const DInst *linst = &sc.inst[light_inst];
  if (linst->kind != MIRT_MAT_EMISSIVE) return V3(0.0f, 0.0f, 0.0f);
  const float *e = sc.emissive + 4 * (size_t)linst->kind_data;
  v3 rad = V3(e[0], e[1], e[2]);
  const uint32_t tex = sc.tri_uv != nullptr ? sc.mat[linst->material].tex[1] : 0u;
  if (tex != 0u) {
    BestHit lb;
    lb.t = 0.0f;
    lb.u = s1.w;
    lb.v = b;
    lb.prim = __float_as_uint(s1.z);
    lb.inst = light_inst;
    float tu, tv;
    hit_uv(sc, lb, tu, tv);
    const v3 c = tex_sample(sc, tex, tu, tv);
    rad = V3(rad.x * c.x, rad.y * c.y, rad.z * c.z);
  }
  return rad;
}

// MirtParams.stage_kernels: the routing pass behind the connect kernel, one thread per entry of the
// connect queue — the slot's ShadowData is completed (shadow[1] = the light sample's radiance) and the slot
// goes to the queue of its hit material.
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
      const float4 *h4 = reinterpret_cast<const float4 *>(st.hit + slot);
      const float4 s0 = sh4[0], s1 = sh4[1];
      v3 rad = V3(0.0f, 0.0f, 0.0f);
      if (s0.w > 0.0f) rad = light_radiance(sc, s1, h4[1].w);
      sh4[1] = make_float4(rad.x, rad.y, rad.z, 0.0f);
      const uint32_t inst_ff = __float_as_uint(h4[0].w);
      const uint32_t kind = sc.inst[inst_ff & 0x7FFFFFFFu].kind;
      // no material (archive/logic.slang:171-174): the path ends here
      route = kind == MIRT_MAT_NONE ? 4 : (int)kind - (int)MIRT_MAT_LAMBERTIAN;
    }
    block_push_n<5>(sh, counts, datas, route, slot);
  }
}

// One shade event of material kind KIND for the slot whose state is in `s` (process_material and the four
// material files of shaders/archive): radiance, throughput and stream state are updated in `s`, the bounce ray
// comes back in new_pos / new_dir.  Returns whether the path goes on.  `use_shadow`: the connect stage ran; `sdw`
// is then the slot's ShadowData — direction to the light sample and its probability (0: none, or not visible),
// `light_rad` the sample's radiance.
template <int KIND>
__device__ __forceinline__ bool shade_material(const DScene &sc, SlotIn &s, bool use_shadow, float4 sdw, v3 light_rad,
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
      const float sprob = sdw.w;
      if (sprob > 0.0f) {
        const v3 sdir = V3(sdw.x, sdw.y, sdw.z);
        float shadow_weight = 1.0f * sprob;
        shadow_weight /= (1.0f * sprob + 1.0f * lambert_pdf(sdir, nf));
        v3 contrib = thr;
        contrib = mul3(contrib, lambert_brdf(albedo, sdir, wo, nf));
        contrib = scale3(contrib, dot3(sdir, nf));
        contrib = scale3(contrib, shadow_weight);
        const float dl = 1.0f * sprob;
        contrib = V3(contrib.x / dl, contrib.y / dl, contrib.z / dl);
        rad = add3(rad, mul3(light_rad, contrib));
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
// MirtParams.stage_kernels — common frame of the four kernels: pop `queue`, shade, push to the next
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
      // the connect stage ran and left ShadowData (only the Lambertian combination reads it)
      float4 sdw = make_float4(0.0f, 0.0f, 0.0f, 0.0f), lrad = sdw;
      if (KIND == MIRT_MAT_LAMBERTIAN) {
        const float4 *sh4 = reinterpret_cast<const float4 *>(st.shadow + slot);
        sdw = sh4[0];
        lrad = sh4[1];
      }
      alive = shade_material<KIND>(sc, s, true, sdw, V3(lrad.x, lrad.y, lrad.z), new_pos, new_dir);
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
// (With light sampling the connect stage sits between the two halves: k_extend_finish, k_connect, k_connect_shade.)
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
  // with enough samples left for every entry, the next queue is this one copied in place (see k_finish_shade_uber)
  const bool all_stay = RESPAWN && st.qcount[F_ALL_STAY] != 0u;
  uint32_t shaded = 0, finished = 0, spawned = 0;  // of this thread: at most one per round of the loop
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
      const uint32_t inst = __float_as_uint(reinterpret_cast<const float4 *>(st.ray + slot)[0].w);  // the walk's result
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
          const float4 no_shadow = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (kind == MIRT_MAT_LAMBERTIAN) alive = shade_material<MIRT_MAT_LAMBERTIAN>(sc, s, false, no_shadow, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
          else if (kind == MIRT_MAT_METALLIC) alive = shade_material<MIRT_MAT_METALLIC>(sc, s, false, no_shadow, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
          else if (kind == MIRT_MAT_DIELECTRIC) alive = shade_material<MIRT_MAT_DIELECTRIC>(sc, s, false, no_shadow, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
          else alive = shade_material<MIRT_MAT_EMISSIVE>(sc, s, false, no_shadow, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
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

// Light sampling, behind the connect kernel, in ONE pass over the connect queue: the epilogue of shadowMain
// (archive/shadow.slang:122-139: the visible sample's radiance) and the hit material's shade event
// (process_material, archive/material.slang:17-139), with the ShadowData in registers between the two — no
// routing pass, no material queues, no second trip of the ShadowData through memory.  A wave whose lanes hit
// different kinds of material runs their branches one after the other, as in k_finish_shade_classic.
//
// RESPAWN: a path that ends here — no material, absorbed by its material — is accumulated and its slot
// takes the render call's next sample at once.
template <bool RESPAWN>
__global__ __launch_bounds__(kStreamBlock, MIRT_MATERIAL_WAVES) void k_connect_shade(DScene sc, DState st, DFrameParams fp) {
  __shared__ BlockPush2 sh;
  __shared__ BlockReserve shr;
  const uint32_t par = fp.frame & 1u;
  const uint32_t *cq = st.qdata[Q_CONNECT];
  const uint32_t count = st.qcount[Q_CONNECT];
  uint32_t shaded = 0, finished = 0, spawned = 0;  // of this thread: at most one per round of the loop
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {  // uniform per block
    const uint32_t i = base + threadIdx.x;
    const bool active = i < count;
    uint32_t slot = 0;
    bool alive = false;
    float4 rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // of a path that ends here
    uint32_t source = kNoSample;
    if (active) {
      slot = cq[i];
      const float4 *sh4 = reinterpret_cast<const float4 *>(st.shadow + slot);
      const float4 sdw = sh4[0], s1 = sh4[1];
      SlotIn s;
      (void)load_slot<false>(sc, st, slot, s);
      const uint32_t kind = sc.inst[s.inst].kind;
      rad = s.pa;
      source = __float_as_uint(s.pb.w);
      if (kind != MIRT_MAT_NONE) {  // no material (archive/logic.slang:171-174): the path ends as it is
        v3 new_pos, new_dir;
        if (kind == MIRT_MAT_LAMBERTIAN) {
          v3 lrad = V3(0.0f, 0.0f, 0.0f);
          if (sdw.w > 0.0f) lrad = light_radiance(sc, s1, s.hit_w);
          alive = shade_material<MIRT_MAT_LAMBERTIAN>(sc, s, true, sdw, lrad, new_pos, new_dir);
        } else if (kind == MIRT_MAT_METALLIC) {
          alive = shade_material<MIRT_MAT_METALLIC>(sc, s, true, sdw, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
        } else if (kind == MIRT_MAT_DIELECTRIC) {
          alive = shade_material<MIRT_MAT_DIELECTRIC>(sc, s, true, sdw, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
        } else {
          alive = shade_material<MIRT_MAT_EMISSIVE>(sc, s, true, sdw, V3(0.0f, 0.0f, 0.0f), new_pos, new_dir);
        }
        shaded++;
        rad = s.pa;
        if (alive || !RESPAWN) store_slot(st, slot, s, new_pos, new_dir);
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
    if (shaded) atomicAdd(&st.stats[S_SHADE_EVENTS], (unsigned long long)shaded);
    if (RESPAWN && finished) atomicAdd(&st.stats[S_FINISHED], (unsigned long long)finished);
    if (RESPAWN && spawned) atomicAdd(&st.stats[S_SPAWNED], (unsigned long long)spawned);
  }
}

}  // namespace

void launch_finish_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  if (fp.respawn) hipLaunchKernelGGL((k_finish_shade_classic<true>), stream_grid(cfg, k_finish_shade_classic<true>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_finish_shade_classic<false>), stream_grid(cfg, k_finish_shade_classic<false>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
}

void launch_extend_finish(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  const dim3 block(kStreamBlock);
  if (fp.respawn) hipLaunchKernelGGL((k_extend_finish<true>), stream_grid(cfg, k_extend_finish<true>), block, 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_extend_finish<false>), stream_grid(cfg, k_extend_finish<false>), block, 0, cfg.stream, sc, st, fp);
}

void launch_connect(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  const dim3 grid(trace_grid_blocks(cfg.num_cus));
  if (fp.counting) hipLaunchKernelGGL((k_connect<true>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_connect<false>), grid, dim3(kEngineBlock), 0, cfg.stream, sc, st, fp);
}

void launch_connect_shade(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  if (fp.respawn) hipLaunchKernelGGL((k_connect_shade<true>), stream_grid(cfg, k_connect_shade<true>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
  else hipLaunchKernelGGL((k_connect_shade<false>), stream_grid(cfg, k_connect_shade<false>), dim3(kStreamBlock), 0, cfg.stream, sc, st, fp);
}

void launch_shade_classic(const LaunchCfg &cfg, const DScene &sc, const DState &st, const DFrameParams &fp) {
  // the routing pass, then the four per-material kernels, one queue each (archive frame order:
  // src/app.rs:402-410 dispatches every material phase each frame)
  const dim3 block(kStreamBlock);
  hipLaunchKernelGGL(k_connect_finish, stream_grid(cfg, k_connect_finish), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_LAMBERTIAN>), stream_grid(cfg, k_material<MIRT_MAT_LAMBERTIAN>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_METALLIC>), stream_grid(cfg, k_material<MIRT_MAT_METALLIC>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_DIELECTRIC>), stream_grid(cfg, k_material<MIRT_MAT_DIELECTRIC>), block, 0, cfg.stream, sc, st, fp);
  hipLaunchKernelGGL((k_material<MIRT_MAT_EMISSIVE>), stream_grid(cfg, k_material<MIRT_MAT_EMISSIVE>), block, 0, cfg.stream, sc, st, fp);
}

}  // namespace mirt
