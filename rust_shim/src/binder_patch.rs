//! binder_patch.rs — the tail of binder_system (src/binder.rs:257-373).  Everything up to
//! `if binder_local.tlas_regenerate` stays: the system still gathers instances, materials,
//! transforms and light_sources from the ECS query.  What changes is what happens to the
//! vectors: instead of ten `create_buffer_init` calls and bind group 0 they are handed to
//! libmirt, which copies them during the call.
//!
//! `SceneBindings` loses its wgpu members:
//!     #[derive(Resource, Default)]
//!     pub struct SceneBindings { pub uploaded: bool }
//! `BinderLocal` keeps `tlas_regenerate` and caches the two TLAS vectors instead of buffers:
//!     tlas_nodes: Vec<BVHNodeGPU>, tlas_iids: Vec<u32>, geometry_generation: u64
//! `MeshServer` keeps its four `Vec`s (vertices, indices, blas_nodes, geometry offsets) next to —
//! or instead of — the wgpu buffers regenerate_buffer builds (src/mesh.rs:198-284) and
//! exposes them as slices; `generation()` counts its regenerations.
use crate::{bvh::BVHNodeGPU, ffi, pathtracer_manager::MirtContext};

#[allow(clippy::too_many_arguments)]
pub fn upload_scene(
    ctx: &MirtContext,
    mesh_server: &crate::mesh::MeshServer,
    instances: &[crate::instance::Instance],
    materials: &[crate::material::Material],
    transforms: &[crate::transform::Transform],
    light_sources: &[u32],
    tlas_nodes: &[BVHNodeGPU],
    tlas_iids: &[u32],
    geometry_changed: bool,
) {
    let desc = ffi::MirtSceneDesc {
        instances: instances.as_ptr(),                        n_instances: instances.len() as u32,
        geometry_offsets: mesh_server.offsets().as_ptr(),     n_geometries: mesh_server.offsets().len() as u32,
        vertices: mesh_server.vertices().as_ptr(),            n_vertices: mesh_server.vertices().len() as u32,
        indices: mesh_server.indices().as_ptr(),              n_indices: mesh_server.indices().len() as u32,
        blas_nodes: mesh_server.blas_nodes().as_ptr(),        n_blas_nodes: mesh_server.blas_nodes().len() as u32,
        materials: materials.as_ptr(),                        n_materials: materials.len() as u32,
        transforms: transforms.as_ptr(),                      n_transforms: transforms.len() as u32,
        tlas_nodes: tlas_nodes.as_ptr(),                      n_tlas_nodes: tlas_nodes.len() as u32,
        tlas_to_instances: tlas_iids.as_ptr(),                n_tlas_to_instances: tlas_iids.len() as u32,
        light_sources: light_sources.as_ptr(),                n_light_sources: light_sources.len() as u32,
        // the per-material kernels and textures are optional: the live pipeline is the uber shader
        classic_refs: std::ptr::null(),                       n_classic_refs: 0,
        lambertian: std::ptr::null(),                         n_lambertian: 0,
        metallic: std::ptr::null(),                           n_metallic: 0,
        dielectric: std::ptr::null(),                         n_dielectric: 0,
        emissive: std::ptr::null(),                           n_emissive: 0,
        textures: std::ptr::null(),                           n_textures: 0,
        texels: std::ptr::null(),                             n_texels: 0,
    };
    // a mesh-server regeneration re-packs nodes and triangles; a moved transform or a changed
    // material only rewrites the TLAS block, the instance matrices and the small tables
    let rc = unsafe {
        if geometry_changed { ffi::mirt_scene_upload(ctx.0, &desc) } else { ffi::mirt_scene_update_instances(ctx.0, &desc) }
    };
    ffi::check(ctx.0, rc, "mirt_scene_upload");
}

/// Only `Transform`s changed since the last upload (the `Changed<Transform>` arm that sets
/// `tlas_regenerate`, src/binder.rs:257-282, with the same instances and materials as before):
/// the TLAS is rebuilt and written in place on the device, `TLAS::new` is not run on the host.
/// Returns false when the new tree does not fit the block reserved at upload; the caller then
/// rebuilds on the host and calls `upload_scene(.., geometry_changed = false)`.
pub fn move_transforms(ctx: &MirtContext, transforms: &[crate::transform::Transform]) -> bool {
    let rc = unsafe {
        ffi::mirt_scene_update_transforms(ctx.0, transforms.as_ptr(), transforms.len() as u32, std::ptr::null_mut())
    };
    rc == ffi::MIRT_OK
}
