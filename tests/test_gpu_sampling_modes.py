"""GPU parity of the two sampling modes beyond the default: the reference's per-slot
random streams (MIRT_RNG_PER_SLOT) and adaptive sampling (converged sources skipped at
spawn).  Everything is compared bit-exact with the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from raytracer_amd import ffi, host, pathtracer
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    p = pathtracer.Pathtracer(0)
    yield p
    p.close()


def host_seeds(slots, seed):
    # PathtracerState::new (src/pathtracer_state.rs:98-110): every word uniform in [1000, 2^32)
    return np.random.default_rng(seed).integers(1000, 2 ** 32, size=(slots, 4), dtype=np.uint64).astype("<u4")


@pytest.mark.parametrize("cfg,W,H,slots", [(1, 64, 64, 64 * 64), (1, 64, 64, 1500), (2, 48, 32, 777)])
def test_per_slot_streams_match_the_oracle_simulation(pt, config_scenes, cfg, W, H, slots):
    arrays, info = config_scenes(cfg)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    rng = host_seeds(slots, 11 + cfg)
    for mode in (0, 1):
        kw = dict(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode, rng_mode=ffi.RNG_PER_SLOT)
        prm, oprm = pathtracer.default_params(**kw), po.default_params(**kw)
        pt.create_state(W, H, slots, src, prm)
        pt.set_camera(cam)
        with pytest.raises(pathtracer.MirtError, match="set_rng"):
            pt.step(1)
        # the per-frame call: F frames of the regenerating wavefront
        frames = 17
        pt.set_rng(rng)
        pt.stats_reset()
        pt.step(frames)
        s4, cnt = pt.read_accum()
        w4, wcnt, wst = sc.wavefront_per_slot(cam, src, W, H, oprm, rng, 2 ** 62, frames, slots)
        assert wst.finished_samples > slots // 2
        assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
        st = pt.stats()
        assert (st.extend_rays, st.connect_rays, st.shade_events, st.finished_samples) == (
            wst.extend_rays, wst.connect_rays, wst.shade_events, wst.finished_samples)
        # a budgeted render from freshly seeded streams: spp samples per source, drained
        pt.reset_accum()
        pt.set_rng(rng)
        pt.render_spp(2)
        s4, cnt = pt.read_accum()
        w4, wcnt, wst = sc.wavefront_per_slot(cam, src, W, H, oprm, rng, 2 * len(src), 2 ** 62, slots)
        assert np.array_equal(cnt, wcnt) and cnt.min() == 2 and np.array_equal(s4, w4)
        # and the picture does depend on the streams: other seeds, other sums
        pt.reset_accum()
        pt.set_rng(host_seeds(slots, 99))
        pt.render_spp(2)
        assert not np.array_equal(pt.read_accum()[0], w4)


@pytest.mark.parametrize("W,H,slots,frames,spp", [(160, 120, 70001, 7, 0), (256, 256, 1024 * 1025 + 3, 3, 0),
                                                  (96, 96, 5000, 0, 3)])
def test_per_slot_sample_stage_over_many_tiles(pt, config_scenes, W, H, slots, frames, spp):
    """The ordered sample stage on slot counts that span many tiles of 1024 slots (one block per tile, the
    tile prefixes from the block that finishes last — with more than 1024 tiles a run of tiles per thread), and
    with a sample budget that runs out in the middle of a tile: sums, counts and statistics of the oracle's
    sequential simulation, where the slots of the terminate queue take sample numbers in ascending slot order."""
    arrays, info = config_scenes(1)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    rng = host_seeds(slots, 5)
    kw = dict(max_bounces=info.max_bounces, seed=info.seed, rng_mode=ffi.RNG_PER_SLOT)
    prm, oprm = pathtracer.default_params(**kw), po.default_params(**kw)
    pt.create_state(W, H, slots, src, prm)
    pt.set_camera(cam)
    pt.set_rng(rng)
    pt.stats_reset()
    if spp == 0:
        pt.step(frames)
        w4, wcnt, wst = sc.wavefront_per_slot(cam, src, W, H, oprm, rng, 2 ** 62, frames, slots)
    else:
        # a render call hands out exactly spp samples per source however many slots ask in its last frames
        # (27 648 samples over 5 000 slots: the budget runs out in the middle of a tile)
        pt.render_spp(spp)
        w4, wcnt, wst = sc.wavefront_per_slot(cam, src, W, H, oprm, rng, spp * len(src), 2 ** 62, slots)
    s4, cnt = pt.read_accum()
    assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
    st = pt.stats()
    assert (st.extend_rays, st.shade_events, st.finished_samples) == (wst.extend_rays, wst.shade_events, wst.finished_samples)
    assert wst.finished_samples > 0


def test_adaptive_sampling_flags_and_skips_like_the_oracle(pt, config_scenes):
    # configuration 1 (suzanne under a light, 256 x 256) with a fixed threshold: rounds of 8
    # samples per still-active source, a convergence pass after each
    arrays, info = config_scenes(1)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W = H = 256
    src = host.make_sample_sources(W, H, info.seed)
    src["flags"][5] = 1  # a source the caller hands over as converged is never sampled
    cam = host.default_camera(W, H)
    kw = dict(max_bounces=info.max_bounces, seed=info.seed, adaptive=1)
    prm, oprm = pathtracer.default_params(**kw), po.default_params(**kw)
    n = len(src)
    pt.create_state(W, H, 20000, src, prm)
    pt.set_camera(cam)
    w4, wcnt, wmom = np.zeros(4 * n, "<u4"), np.zeros(n, "<u4"), np.zeros(2 * n, "<u8")
    wflags = (src["flags"] & 1).astype("<u4")
    first, thr, min_samples = 0, np.float32(0.08), 8
    left_history = []
    for round_no in range(4):
        active = np.flatnonzero(wflags == 0).astype("<u4")
        pt.render_spp(8)
        sc.render_active(cam, src, W, H, oprm, first, 8, active, w4, wcnt, wmom)
        first += 8
        s4, cnt = pt.read_accum()
        assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
        assert np.array_equal(pt.read_moments(), wmom)
        left = pt.adapt_update(thr, min_samples)
        assert left == po.adapt_update(wcnt, wmom, wflags, thr, min_samples)
        assert np.array_equal(pt.read_flags(), wflags)
        left_history.append(left)
    assert wcnt[5] == 0 and wflags[5] == 1
    # the sky converges at once (every sample is the same), suzanne's silhouette and shadow do not
    assert left_history[0] < n // 2 and left_history[-1] < left_history[0] and left_history[-1] > 0
    assert wcnt.max() == 32 and np.count_nonzero(wcnt == 8) > n // 2
    # moments are integers: two devices' worth add up to one device's (sample ranges 0..3 and 4..7)
    halves = []
    for first_sample in (0, 4):
        pt.reset_accum()
        pt.set_sample_range(first_sample)
        pt.render_spp(4)
        halves.append(pt.read_moments())
    assert wcnt_sum_equal(halves, sc, cam, src, W, H, oprm)
    # a camera change clears flags and moments (sampleCleanup, shaders/sample.slang:123-145)
    pt.adapt_update(thr, 2)
    assert pt.read_flags().any()
    pt.reset_accum()
    assert not pt.read_flags().any() and not pt.read_moments().any()
    with pytest.raises(pathtracer.MirtError, match="adaptive"):
        pt.create_state(W, H, 100, src, pathtracer.default_params(max_bounces=4))
        pt.adapt_update(thr, 2)


def wcnt_sum_equal(halves, sc, cam, src, W, H, oprm):
    n = len(src)
    w4, wcnt, wmom = np.zeros(4 * n, "<u4"), np.zeros(n, "<u4"), np.zeros(2 * n, "<u8")
    sc.render_active(cam, src, W, H, oprm, 0, 8, None, w4, wcnt, wmom)
    return np.array_equal(halves[0] + halves[1], wmom)


def test_group_adaptive_rounds_flag_the_same_sources_on_every_member(config_scenes):
    """mirt_group_adapt_update on two contexts (both on device 0 here): every round's samples are split over the
    members, the reduction brings sums, counts and moments together on member 0, which flags from the totals, and
    every member takes its flags — round after round the same accumulators, moments, flags and active count as one
    context rendering all the samples, and (first round) as the oracle."""
    arrays, info = config_scenes(1)
    W = H = 128
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    kw = dict(max_bounces=info.max_bounces, seed=info.seed, adaptive=1)
    prm, oprm = pathtracer.default_params(**kw), po.default_params(**kw)
    thr, min_samples = np.float32(0.08), 8
    n = len(src)
    with pathtracer.Pathtracer(0) as one, pathtracer.PathtracerGroup([0, 0]) as grp:
        for p in (one, grp):
            p.upload_scene(arrays)
            p.create_state(W, H, 6000, src, prm)
            p.set_camera(cam)
        history = []
        for round_no in range(3):
            one.render_spp(8)
            grp.render_spp(8)
            left = one.adapt_update(thr, min_samples)
            assert grp.adapt_update(thr, min_samples) == left
            history.append(left)
            s4, cnt = one.read_accum()
            g4, gcnt = grp.read_accum()
            assert np.array_equal(g4, s4) and np.array_equal(gcnt, cnt)
            assert np.array_equal(grp.members[0].read_moments(), one.read_moments())
            flags = one.read_flags()
            for m in grp.members:
                assert np.array_equal(m.read_flags(), flags)
            for m in grp.members[1:]:  # cleared by the reduction, moments too
                assert not m.read_accum()[1].any() and not m.read_moments().any()
            if round_no == 0:
                w4, wcnt, wmom = np.zeros(4 * n, "<u4"), np.zeros(n, "<u4"), np.zeros(2 * n, "<u8")
                wflags = np.zeros(n, "<u4")
                po.Scene(arrays).render_active(cam, src, W, H, oprm, 0, 8, np.arange(n, dtype="<u4"), w4, wcnt, wmom)
                assert np.array_equal(g4, w4) and np.array_equal(grp.members[0].read_moments(), wmom)
                assert po.adapt_update(wcnt, wmom, wflags, thr, min_samples) == left and np.array_equal(flags, wflags)
        assert 0 < history[-1] < history[0] < n
