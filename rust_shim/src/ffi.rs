//! ffi.rs — the C ABI of libmirt (include/mirt.h), declaration by declaration.
//! The scene PODs are the crate's own `#[repr(C)]` bytemuck structs
//! (Instance, GeometryOffsets, GPUVertexData, BVHNodeGPU, Material, Transform,
//! SampleSource, CameraData): libmirt reads exactly those bytes.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int};

#[repr(C)] pub struct mirt_ctx { _private: [u8; 0] }
#[repr(C)] pub struct mirt_group { _private: [u8; 0] }

pub const MIRT_OK: c_int = 0;
pub const MIRT_SHADE_UBER: u32 = 0;
pub const MIRT_SHADE_CLASSIC: u32 = 1;
pub const MIRT_RNG_PER_SAMPLE: u32 = 0;
pub const MIRT_RNG_PER_SLOT: u32 = 1;
pub const MIRT_FILTER_NEAREST: u32 = 0;
pub const MIRT_FILTER_BILINEAR: u32 = 1;

#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtClassicMaterialRef { pub kind: u32, pub data: u32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtTexture { pub width: u32, pub height: u32, pub first_texel: u32, pub filter: u32 }

#[repr(C)]
pub struct MirtSceneDesc {
    pub instances: *const crate::instance::Instance,           pub n_instances: u32,
    pub geometry_offsets: *const crate::mesh::GeometryOffsets, pub n_geometries: u32,
    pub vertices: *const crate::mesh::GPUVertexData,           pub n_vertices: u32,
    pub indices: *const glam::UVec4,                           pub n_indices: u32,
    pub blas_nodes: *const crate::bvh::BVHNodeGPU,             pub n_blas_nodes: u32,
    pub materials: *const crate::material::Material,           pub n_materials: u32,
    pub transforms: *const crate::transform::Transform,        pub n_transforms: u32,
    pub tlas_nodes: *const crate::bvh::BVHNodeGPU,             pub n_tlas_nodes: u32,
    pub tlas_to_instances: *const u32,                         pub n_tlas_to_instances: u32,
    pub light_sources: *const u32,                             pub n_light_sources: u32,
    pub classic_refs: *const MirtClassicMaterialRef,           pub n_classic_refs: u32,
    pub lambertian: *const crate::lambertian::LambertianData,  pub n_lambertian: u32,
    pub metallic: *const crate::metallic::MetallicData,        pub n_metallic: u32,
    pub dielectric: *const crate::dielectric::DielectricData,  pub n_dielectric: u32,
    pub emissive: *const crate::emissive::EmissiveData,        pub n_emissive: u32,
    pub textures: *const MirtTexture,                          pub n_textures: u32,
    pub texels: *const u32,                                    pub n_texels: u32,
}

#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtParams {
    pub max_bounces: u32,
    pub sky_radiance: f32,
    pub fixed_point_scale: f32,
    pub exposure: f32,
    pub shade_mode: u32,
    pub rng_mode: u32,
    pub seed: u32,
    pub nee: u32,
    pub adaptive: u32,
    pub stage_kernels: u32,
}

#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtRay { pub pos: [f32; 3], pub dir: [f32; 3], pub last_inst: u32, pub last_prim: u32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtHitId { pub instance_id: u32, pub triangle_id: u32, pub t: f32, pub front_face: u32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtTraversalCounts {
    pub rays: u64,
    pub tlas_nodes: u64,
    pub instance_visits: u64,
    pub blas_nodes: u64,
    pub triangles: u64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtStats {
    pub extend_rays: u64,
    pub connect_rays: u64,
    pub shade_events: u64,
    pub spawned_samples: u64,
    pub finished_samples: u64,
    pub frames: u64,
    pub extend_launches: u64,   pub extend_ms: f64,
    pub finish_launches: u64,   pub finish_ms: f64,
    pub connect_launches: u64,  pub connect_ms: f64,
    pub shade_launches: u64,    pub shade_ms: f64,
    pub sample_launches: u64,   pub sample_ms: f64,
    pub extend_counts: MirtTraversalCounts,
    pub connect_counts: MirtTraversalCounts,
    pub extend_tri_wave_lds: u64,  pub extend_tri_wave_mem: u64,
    pub connect_tri_wave_lds: u64, pub connect_tri_wave_mem: u64,
    pub overflowed_sums: u64,
    pub engine_phase: [u64; 16],
}

extern "C" {
    // lifecycle
    pub fn mirt_create(device_ordinal: c_int, out_ctx: *mut *mut mirt_ctx) -> c_int;
    pub fn mirt_destroy(ctx: *mut mirt_ctx) -> c_int;
    pub fn mirt_last_error(ctx: *const mirt_ctx) -> *const c_char;
    pub fn mirt_status_string(status: c_int) -> *const c_char;
    pub fn mirt_default_params(out: *mut MirtParams);
    // scene and state
    pub fn mirt_scene_upload(ctx: *mut mirt_ctx, scene: *const MirtSceneDesc) -> c_int;
    pub fn mirt_scene_update_instances(ctx: *mut mirt_ctx, scene: *const MirtSceneDesc) -> c_int;
    pub fn mirt_scene_update_transforms(ctx: *mut mirt_ctx, transforms: *const crate::transform::Transform,
                                        n_transforms: u32, device_ms: *mut f32) -> c_int;
    pub fn mirt_state_create(ctx: *mut mirt_ctx, width: u32, height: u32, threads: u32,
                             sources: *const crate::pathtracer_state::SampleSource, n_sources: u32,
                             params: *const MirtParams) -> c_int;
    pub fn mirt_state_set_rng(ctx: *mut mirt_ctx, rng: *const [u32; 4], n_slots: u32) -> c_int;
    pub fn mirt_camera_set(ctx: *mut mirt_ctx, camera: *const crate::camera::CameraData) -> c_int;
    pub fn mirt_reset_accum(ctx: *mut mirt_ctx) -> c_int;
    pub fn mirt_set_sample_range(ctx: *mut mirt_ctx, first_sample: u32) -> c_int;
    // rendering
    pub fn mirt_step(ctx: *mut mirt_ctx, n_frames: u32) -> c_int;
    pub fn mirt_render_spp(ctx: *mut mirt_ctx, spp: u32) -> c_int;
    pub fn mirt_synchronize(ctx: *mut mirt_ctx) -> c_int;
    pub fn mirt_read_output(ctx: *mut mirt_ctx, rgb: *mut u32, n_pixels: usize) -> c_int;
    pub fn mirt_read_accum(ctx: *mut mirt_ctx, sum4: *mut u32, count: *mut u32, n_sources: usize) -> c_int;
    // adaptive sampling
    pub fn mirt_adapt_update(ctx: *mut mirt_ctx, rel_threshold: f32, min_samples: u32, n_active: *mut u32) -> c_int;
    pub fn mirt_read_flags(ctx: *mut mirt_ctx, flags: *mut u32, n_sources: usize) -> c_int;
    pub fn mirt_read_moments(ctx: *mut mirt_ctx, moments: *mut u64, n_sources: usize) -> c_int;
    pub fn mirt_write_moments(ctx: *mut mirt_ctx, moments: *const u64, n_sources: usize) -> c_int;
    // accumulators in device memory (for a collective driven by the caller)
    pub fn mirt_copy_accum_to_device(ctx: *mut mirt_ctx, dev_sum4: *mut std::ffi::c_void,
                                     dev_count: *mut std::ffi::c_void, n_sources: usize) -> c_int;
    pub fn mirt_load_accum_from_device(ctx: *mut mirt_ctx, dev_sum4: *const std::ffi::c_void,
                                       dev_count: *const std::ffi::c_void, n_sources: usize) -> c_int;
    // ray sets
    pub fn mirt_trace_closest(ctx: *mut mirt_ctx, rays: *const MirtRay, n: usize, out: *mut MirtHitId,
                              counts: *mut MirtTraversalCounts) -> c_int;
    pub fn mirt_trace_closest_static(ctx: *mut mirt_ctx, rays: *const MirtRay, n: usize, out: *mut MirtHitId,
                                     counts: *mut MirtTraversalCounts) -> c_int;
    pub fn mirt_trace_any(ctx: *mut mirt_ctx, rays: *const MirtRay, tmax: *const f32, n: usize,
                          out: *mut MirtHitId, counts: *mut MirtTraversalCounts) -> c_int;
    pub fn mirt_bench_trace_closest(ctx: *mut mirt_ctx, rays: *const MirtRay, n: usize, repeats: u32,
                                    ms_per_launch: *mut f64) -> c_int;
    pub fn mirt_bench_gather(ctx: *mut mirt_ctx, table_bytes: usize, steps: u32, repeats: u32,
                             lane_steps_per_s: *mut f64) -> c_int;
    pub fn mirt_scene_bytes(ctx: *mut mirt_ctx, node_and_triangle_bytes: *mut u64, leaf_cache_bytes: *mut u64) -> c_int;
    // BLAS::new on the device
    pub fn mirt_build_blas(ctx: *mut mirt_ctx, positions4: *const f32, n_positions: u32, faces4: *mut u32,
                           n_faces: u32, nodes: *mut crate::bvh::BVHNodeGPU, nodes_capacity: u32,
                           n_nodes: *mut u32, device_ms: *mut f32) -> c_int;
    // TLAS::new on the device
    pub fn mirt_build_tlas(ctx: *mut mirt_ctx, geom_aabbs6: *const f32, n_geometries: u32,
                           transforms: *const crate::transform::Transform, n_transforms: u32,
                           instances: *const crate::instance::Instance, n_instances: u32,
                           nodes: *mut crate::bvh::BVHNodeGPU, nodes_capacity: u32, n_nodes: *mut u32,
                           instance_ids: *mut u32, device_ms: *mut f32) -> c_int;
    // several GPUs from this one process
    pub fn mirt_group_create(device_ordinals: *const c_int, n: c_int, out_group: *mut *mut mirt_group) -> c_int;
    pub fn mirt_group_destroy(grp: *mut mirt_group) -> c_int;
    pub fn mirt_group_size(grp: *const mirt_group) -> c_int;
    pub fn mirt_group_member(grp: *mut mirt_group, i: c_int) -> *mut mirt_ctx;
    pub fn mirt_group_last_error(grp: *const mirt_group) -> *const c_char;
    pub fn mirt_group_scene_upload(grp: *mut mirt_group, scene: *const MirtSceneDesc) -> c_int;
    pub fn mirt_group_scene_update_instances(grp: *mut mirt_group, scene: *const MirtSceneDesc) -> c_int;
    pub fn mirt_group_state_create(grp: *mut mirt_group, width: u32, height: u32, threads: u32,
                                   sources: *const crate::pathtracer_state::SampleSource, n_sources: u32,
                                   params: *const MirtParams) -> c_int;
    pub fn mirt_group_camera_set(grp: *mut mirt_group, camera: *const crate::camera::CameraData) -> c_int;
    pub fn mirt_group_reset_accum(grp: *mut mirt_group) -> c_int;
    pub fn mirt_group_render_spp(grp: *mut mirt_group, spp: u32) -> c_int;
    pub fn mirt_group_reduce(grp: *mut mirt_group) -> c_int;
    pub fn mirt_group_adapt_update(grp: *mut mirt_group, rel_threshold: f32, min_samples: u32, n_active: *mut u32) -> c_int;
    pub fn mirt_group_scene_update_transforms(grp: *mut mirt_group, transforms: *const crate::transform::Transform,
                                              n_transforms: u32, device_ms: *mut f32) -> c_int;
    // diagnostics
    pub fn mirt_stats(ctx: *mut mirt_ctx, out: *mut MirtStats) -> c_int;
    pub fn mirt_stats_reset(ctx: *mut mirt_ctx) -> c_int;
    pub fn mirt_set_counting(ctx: *mut mirt_ctx, enabled: c_int) -> c_int;
    pub fn mirt_set_event_timing(ctx: *mut mirt_ctx, enabled: c_int) -> c_int;
    pub fn mirt_dump_extension_rays(ctx: *mut mirt_ctx, out: *mut MirtRay, capacity: usize, n_written: *mut usize) -> c_int;
    pub fn mirt_math_selftest(ctx: *mut mirt_ctx, a: *const f32, b: *const f32, c: *const f32, n: usize,
                              out6: *mut f32) -> c_int;
}

/// The message of the last failure on `ctx`.
pub fn last_error(ctx: *const mirt_ctx) -> String {
    unsafe { std::ffi::CStr::from_ptr(mirt_last_error(ctx)).to_string_lossy().into_owned() }
}

/// Panics with the library's message, like the reference's `unwrap()`s do.
pub fn check(ctx: *const mirt_ctx, rc: c_int, what: &str) {
    assert!(rc == MIRT_OK, "{what}: {}", last_error(ctx));
}
