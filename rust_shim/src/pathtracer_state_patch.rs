//! pathtracer_state_patch.rs — what is left of PathtracerState (src/pathtracer_state.rs:73-353):
//! the two host-side vectors its constructor generates and used to upload at once.  The
//! 19 wgpu bindings, the four queues and the bind group live inside libmirt now.
use bevy_ecs::prelude::*;
use itertools::Itertools;
use rand::{Rng, seq::SliceRandom};

use crate::pathtracer_state::SampleSource;

#[derive(Component)]
pub struct PathtracerState {
    /// one uint4 per path slot, every word in 1000..=u32::MAX (was: random_state_buffer, :98-110)
    pub random_states: Vec<[u32; 4]>,
    /// tile-shuffled sample sources (was: sampling_source_buffer, :175-201)
    pub sample_sources: Vec<SampleSource>,
}

impl PathtracerState {
    pub fn new(dims: (u32, u32), threads: u32) -> Self {
        let mut rng = rand::rng();
        let random_states = (0..threads)
            .map(|_| {
                [
                    rng.random_range(1000..=u32::MAX),
                    rng.random_range(1000..=u32::MAX),
                    rng.random_range(1000..=u32::MAX),
                    rng.random_range(1000..=u32::MAX),
                ]
            })
            .collect();
        let tile_size = 128;
        // div_ceil and the min() below cover the pixels the reference's whole-tile loop leaves out
        // when a dimension is not a multiple of 128 (1080 rows: 1024..1079)
        let mut tiles = (0..dims.0.div_ceil(tile_size)).cartesian_product(0..dims.1.div_ceil(tile_size)).collect_vec();
        tiles.shuffle(&mut rng);
        let mut sample_sources = tiles
            .into_iter()
            .flat_map(|(x, y)| {
                ((x * tile_size)..(x * tile_size + tile_size).min(dims.0))
                    .cartesian_product((y * tile_size)..(y * tile_size + tile_size).min(dims.1))
                    .map(|(x, y)| SampleSource {
                        screen_pos: [x as f32 / dims.0 as f32, y as f32 / dims.1 as f32],
                        out_pos: [x, y],
                        samples: 0,
                        flags: 0,
                    })
            })
            .collect_vec();
        sample_sources.shuffle(&mut rng);
        Self { random_states, sample_sources }
    }
}
