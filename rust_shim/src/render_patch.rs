//! render_patch.rs — replaces PathtracerOutput::copy_to_texture (src/pathtracer.rs:124-146):
//! the packed pixels come back from libmirt and go into the same Rgba8UnormSrgb texture
//! the present pass samples (bytes R, G, B, 0, like the reference's packRgb).
use crate::{ffi, pathtracer::PathtracerOutput, pathtracer_manager::MirtContext};

impl PathtracerOutput {
    pub fn copy_to_texture(&self, ctx: &MirtContext, queue: &wgpu::Queue, pixels: &mut Vec<u32>) {
        let size = self.out_texture.size();
        pixels.resize((size.width * size.height) as usize, 0);
        // tone-maps the current means and synchronises with the frames submitted so far
        let rc = unsafe { ffi::mirt_read_output(ctx.0, pixels.as_mut_ptr(), pixels.len()) };
        ffi::check(ctx.0, rc, "mirt_read_output");
        queue.write_texture(
            wgpu::TexelCopyTextureInfoBase {
                texture: &self.out_texture,
                mip_level: 0,
                origin: wgpu::Origin3d::ZERO,
                aspect: wgpu::TextureAspect::All,
            },
            bytemuck::cast_slice(pixels),
            wgpu::TexelCopyBufferLayout {
                offset: 0,
                bytes_per_row: Some(4 * size.width),
                rows_per_image: Some(size.height),
            },
            size,
        );
    }
}
