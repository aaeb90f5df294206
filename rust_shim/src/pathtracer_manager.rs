//! pathtracer_manager.rs — replaces src/pathtracer_manager.rs of the reference.
//! The four wgpu compute pipelines (`PathtracerPhase`, :16-22, :195-281) and the four
//! dispatches of `pathtracer_phase_execute` (:74-124) become calls into libmirt; the ECS
//! scheduling (:24-37) stays as it was.
use bevy_ecs::prelude::*;

use crate::{
    app::BevyApp,
    binder::{SceneBindings, binder_system},
    camera::Camera,
    ffi,
    pathtracer::{Pathtracer, PathtracerOutput, pathtracer_output_sync_system},
    pathtracer_state::PathtracerState,
    render::render_system,
    schedule,
};

/// The HIP context that stands where RenderDevice / RenderQueue stood for the compute
/// passes (src/render_resources.rs:37-116).  One context, used by one system at a time:
/// bevy's `ResMut` access gives exactly that.
#[derive(Resource)]
pub struct MirtContext(pub *mut ffi::mirt_ctx);
// the library takes calls from any thread, one at a time per context
unsafe impl Send for MirtContext {}
unsafe impl Sync for MirtContext {}

impl Drop for MirtContext {
    fn drop(&mut self) {
        unsafe { ffi::mirt_destroy(self.0) };
    }
}

pub fn initialize(app: &mut BevyApp) {
    let mut ctx = std::ptr::null_mut();
    let rc = unsafe { ffi::mirt_create(0, &mut ctx) };
    assert!(rc == ffi::MIRT_OK, "mirt_create: no MI355X device (status {rc})");
    app.world.insert_resource(MirtContext(ctx));
    app.world.get_resource_or_init::<Schedules>().add_systems(
        schedule::Update,
        (
            pathtracer_phase_execute.before(render_system).after(binder_system),
            pathtracer_phase_sync
                .before(pathtracer_phase_execute)
                .after(pathtracer_output_sync_system)
                .after(binder_system),
        ),
    );
}

/// was: PathtracerState::new + PathtracerPhase::new on Changed<PathtracerOutput> (:39-72).
/// The device-side state (path slots, queues, accumulators, output) is created inside the
/// library from the host's sample sources and random states.
fn pathtracer_phase_sync(
    query: Query<(&Pathtracer, &PathtracerState, &Camera), Changed<PathtracerOutput>>,
    ctx: ResMut<MirtContext>,
) {
    for (pt, pts, camera) in query {
        let mut params = unsafe { std::mem::zeroed::<ffi::MirtParams>() };
        unsafe { ffi::mirt_default_params(&mut params) }; // the shader literals: 128 bounces, sky 10, x1000, 2^-2.5
        // the reference's own scheme: one persistent random stream per path slot
        params.rng_mode = ffi::MIRT_RNG_PER_SLOT;
        let rc = unsafe {
            ffi::mirt_state_create(ctx.0, pt.dims.0, pt.dims.1, pt.threads, pts.sample_sources.as_ptr(),
                                   pts.sample_sources.len() as u32, &params)
        };
        ffi::check(ctx.0, rc, "mirt_state_create");
        let rc = unsafe { ffi::mirt_state_set_rng(ctx.0, pts.random_states.as_ptr(), pt.threads) };
        ffi::check(ctx.0, rc, "mirt_state_set_rng");
        let rc = unsafe { ffi::mirt_camera_set(ctx.0, &camera.data) };
        ffi::check(ctx.0, rc, "mirt_camera_set");
    }
}

/// was: one compute pass of sampleCleanup, sampleMain, ray_extend, shade (:74-124).
fn pathtracer_phase_execute(
    query: Query<(&Pathtracer, &Camera)>,
    ctx: ResMut<MirtContext>,
    scene_bindings: Res<SceneBindings>,
) {
    if !scene_bindings.uploaded {
        return; // was: bind_group.is_none()
    }
    for (_pt, camera) in query {
        if camera.data.changed != 0 {
            // sampleCleanup (shaders/sample.slang:123-145): a changed camera restarts the accumulation
            let rc = unsafe { ffi::mirt_camera_set(ctx.0, &camera.data) };
            ffi::check(ctx.0, rc, "mirt_camera_set");
        }
        // one frame: every in-flight path advances by one bounce; returns before the GPU is done
        let rc = unsafe { ffi::mirt_step(ctx.0, 1) };
        ffi::check(ctx.0, rc, "mirt_step");
    }
}
