"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs, against the committed golden vectors, and —
at full size — through size-independent properties.

Bars: hit instance / primitive ids, t bits, front_face, traversal visit counts,
fixed-point radiance sums, sample counts and tone-mapped pixels are all compared
BIT-EXACT (tolerance 0): the kernels and the oracle share one arithmetic
contract (f32, no implicit contraction, correctly rounded / and sqrt).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, hits_equal, random_rays
from raytracer_amd import ffi, host, pathtracer
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(GOLDEN, "golden.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def pt():
    p = pathtracer.Pathtracer(0)
    yield p
    p.close()


def test_arithmetic_contract_bits(pt):
    rng = np.random.default_rng(1)
    n = 200000
    a = (rng.normal(size=n) * 10.0 ** rng.uniform(-20, 20, n)).astype("<f4")
    b = (rng.normal(size=n) * 10.0 ** rng.uniform(-20, 20, n)).astype("<f4")
    c = (rng.normal(size=n) * 10.0 ** rng.uniform(-20, 20, n)).astype("<f4")
    # include denormals, zeros and infinities
    a[:8] = [0.0, -0.0, 1e-42, -1e-42, np.inf, 1.0, 3.0, 1e-38]
    b[:8] = [1.0, 2.0, 3.0, 1e-3, 2.0, 3.0, 1e-42, 1e-7]
    assert np.array_equal(pt.math_selftest(a, b, c).view("<u4"), po.math_selftest(a, b, c).view("<u4"))


@pytest.mark.parametrize("cfg", [1, 2])
def test_golden_ray_sets(pt, config_scenes, cfg):
    arrays, _ = config_scenes(cfg)
    pt.upload_scene(arrays)
    for bounce in (0, 1):
        rays = np.ascontiguousarray(GOLD[f"cfg{cfg}_bounce{bounce}_rays"]).view(ffi.Ray).reshape(-1)
        want = np.ascontiguousarray(GOLD[f"cfg{cfg}_bounce{bounce}_hits"]).view(ffi.HitId).reshape(-1)
        assert hits_equal(pt.trace_closest(rays), want)


@pytest.mark.parametrize("scene_name", ["cfg1", "cfg2", "knot", "cfg4_small"])
def test_closest_and_any_hit_ids_and_visit_counts(pt, config_scenes, small_knot_scene, scene_name):
    if scene_name == "knot":
        arrays = small_knot_scene
        centre = (0.0, 0.0, 3.0)
    elif scene_name == "cfg4_small":
        arrays, _ = config_scenes(4, dragon_n_u=48, dragon_n_v=12)  # 1000 instances, small dragon stand-in
        centre = (0.0, 0.0, 14.0)
    else:
        arrays, _ = config_scenes(int(scene_name[3:]))
        centre = (0.0, 0.0, 3.0)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    rays = random_rays(100003, 21, centre=centre, spread=2.0 if scene_name != "cfg4_small" else 10.0)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want)
    assert gc.as_dict() == wc.as_dict()
    assert (want["instance_id"] != ffi.NO_HIT).sum() > 1000
    # exclusion of the primitive just hit, as the bounce rays use it
    hit = want["instance_id"] != ffi.NO_HIT
    again = rays.copy()
    again["last_inst"] = np.where(hit, want["instance_id"], ffi.NO_PRIM)
    again["last_prim"] = np.where(hit, want["triangle_id"], ffi.NO_PRIM)
    assert hits_equal(pt.trace_closest(again), sc.trace_closest(again))
    # shadow / connect query: first accepted triangle in traversal order
    tmax = np.random.default_rng(3).uniform(0.2, 8.0, len(rays)).astype("<f4")
    wany, wac = sc.trace_any(rays, tmax, True)
    gany, gac = pt.trace_any(rays, tmax, True)
    assert hits_equal(gany, wany)
    assert gac.as_dict() == wac.as_dict()


def test_ray_lists_large_enough_to_be_staged(pt, config_scenes):
    # ray lists of 16 x 128 entries per resident wave and more (8.4 M on 256 CUs) go through the form of the walk
    # that fetches its rays 64 at a time ahead of need, with the list index as the queue entry: closest hits with
    # visit counts and shadow queries against the oracle, on a list that does not end on a chunk
    arrays, _ = config_scenes(1)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    n = 2 * 16 * 128 * 16 * 256 // 2 + 64 * 100000 + 37
    assert n > 8400000 * 1.5
    rays = random_rays(n, 99, centre=(0.0, 0.0, 3.0), spread=2.0)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want)
    assert gc.as_dict() == wc.as_dict()
    assert (want["instance_id"] != ffi.NO_HIT).sum() > 100000
    tmax = np.random.default_rng(4).uniform(0.2, 8.0, n).astype("<f4")
    wany, wac = sc.trace_any(rays, tmax, True)
    gany, gac = pt.trace_any(rays, tmax, True)
    assert hits_equal(gany, wany)
    assert gac.as_dict() == wac.as_dict()


def test_empty_and_ragged_ray_sets(pt, config_scenes):
    arrays, _ = config_scenes(1)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    assert len(pt.trace_closest(np.zeros(0, ffi.Ray))) == 0
    for n in (1, 63, 64, 65, 257):
        rays = random_rays(n, n)
        assert hits_equal(pt.trace_closest(rays), sc.trace_closest(rays))
    # axis-parallel rays (zero direction components -> infinities in the slab test)
    rays = random_rays(512, 77)
    rays["dir"] = 0.0
    rays["dir"][:, 2] = 1.0
    rays["dir"][::2, 2] = -1.0
    rays["pos"][:, 2] = np.where(rays["dir"][:, 2] > 0, -1.0, 8.0)
    assert hits_equal(pt.trace_closest(rays), sc.trace_closest(rays))


@pytest.mark.parametrize("cfg,W,H,spp", [(1, 64, 64, 2), (2, 64, 36, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_render_sums(pt, config_scenes, cfg, W, H, spp, mode):
    # mode 0: uber shading (shade.slang); mode 1: per-material kernels + connect rays
    arrays, info = config_scenes(cfg)
    pt.upload_scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode)
    pt.create_state(W, H, 1000, src, prm)  # far fewer slots than samples: slots are recycled
    pt.set_camera(host.default_camera(W, H))
    pt.stats_reset()
    pt.render_spp(spp)
    s4, cnt = pt.read_accum()
    assert np.array_equal(s4, GOLD[f"cfg{cfg}_mode{mode}_sum4"])
    assert np.array_equal(cnt, GOLD[f"cfg{cfg}_mode{mode}_count"])
    st = pt.stats()
    assert [st.extend_rays, st.connect_rays, st.shade_events] == GOLD[f"cfg{cfg}_mode{mode}_rays"].tolist()
    assert st.finished_samples == W * H * spp == st.spawned_samples


def test_per_material_kernels_and_connect_rays_config5_small(pt, config_scenes):
    # configuration 5's scene (Cornell box with ceiling light + dielectric dragon
    # stand-in), per-material shading with light sampling, against the oracle
    arrays, info = config_scenes(5, dragon_n_u=96, dragon_n_v=24)
    assert info.shade_mode == 1 and info.nee == 1
    kinds = sorted(set(arrays["classic_refs"]["kind"].tolist()))
    assert kinds == [ffi.MAT_LAMBERTIAN, ffi.MAT_DIELECTRIC, ffi.MAT_EMISSIVE]
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = 192, 108
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=8, seed=info.seed, shade_mode=1, nee=1)
    oprm = po.default_params(max_bounces=8, seed=info.seed, shade_mode=1, nee=1)
    pt.create_state(W, H, 5000, src, prm)
    pt.set_camera(cam)
    pt.set_counting(True)
    pt.stats_reset()
    pt.render_spp(3)
    pt.set_counting(False)
    s4, cnt = pt.read_accum()
    w4, wcnt, wst = sc.render(cam, src, W, H, oprm, 0, 3)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
    st = pt.stats()
    assert (st.extend_rays, st.connect_rays, st.shade_events) == (wst.extend_rays, wst.connect_rays, wst.shade_events)
    assert st.connect_rays > 1000
    assert st.extend_counts.as_dict() == wst.extend_counts.as_dict()
    assert st.connect_counts.as_dict() == wst.connect_counts.as_dict()
    # the metallic kernel too (configuration 2 carries a metallic teapot)
    arrays2, info2 = config_scenes(2)
    assert ffi.MAT_METALLIC in arrays2["classic_refs"]["kind"].tolist()


def test_one_kernel_per_material_behind_the_connect_stage(pt, config_scenes):
    # MirtParams.stage_kernels: the archive frame's shape — a routing pass and one kernel per material, each
    # popping its own queue (src/app.rs:402-410) — against the oracle, like the one-kernel default: a budgeted
    # render, the per-frame call, and the per-frame call with per-slot streams (where the order in which a
    # slot's stream is drawn from matters: light sample first, then the material)
    for cfg, W, H, slots in ((5, 128, 72, 3000), (1, 64, 64, 1500)):
        arrays, info = config_scenes(cfg, dragon_n_u=96, dragon_n_v=24) if cfg == 5 else config_scenes(cfg)
        sc = po.Scene(arrays)
        pt.upload_scene(arrays)
        src = host.make_sample_sources(W, H, info.seed)
        cam = host.default_camera(W, H)
        kw = dict(max_bounces=info.max_bounces, seed=info.seed, shade_mode=1, nee=1)
        oprm = po.default_params(**kw)
        w4, wcnt, wst = sc.render(cam, src, W, H, oprm, 0, 2)
        f4, fcnt, fst = sc.step_wavefront(cam, src, W, H, oprm, 0, 11, slots)
        assert wst.connect_rays > 300 and fst.connect_rays > 300 and fst.finished_samples > 0
        for stage in (1, 0):
            pt.create_state(W, H, slots, src, pathtracer.default_params(stage_kernels=stage, **kw))
            pt.set_camera(cam)
            pt.stats_reset()
            pt.render_spp(2)
            s4, cnt = pt.read_accum()
            assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4), (cfg, stage)
            st = pt.stats()
            assert (st.extend_rays, st.connect_rays, st.shade_events) == (wst.extend_rays, wst.connect_rays, wst.shade_events)
            pt.reset_accum()
            pt.create_state(W, H, slots, src, pathtracer.default_params(stage_kernels=stage, **kw))
            pt.set_camera(cam)
            pt.stats_reset()
            pt.step(11)
            s4, cnt = pt.read_accum()
            assert np.array_equal(cnt, fcnt) and np.array_equal(s4, f4), (cfg, stage)
            st = pt.stats()
            assert (st.extend_rays, st.connect_rays, st.shade_events, st.finished_samples) == (
                fst.extend_rays, fst.connect_rays, fst.shade_events, fst.finished_samples)
        # per-slot streams
        rng = np.random.default_rng(5 + cfg).integers(1000, 2 ** 32, size=(slots, 4), dtype=np.uint64).astype("<u4")
        okw = dict(rng_mode=ffi.RNG_PER_SLOT, **kw)
        p4, pcnt, pst = sc.wavefront_per_slot(cam, src, W, H, po.default_params(**okw), rng, 2 ** 62, 13, slots)
        for stage in (1, 0):
            pt.create_state(W, H, slots, src, pathtracer.default_params(stage_kernels=stage, **okw))
            pt.set_camera(cam)
            pt.set_rng(rng)
            pt.stats_reset()
            pt.step(13)
            s4, cnt = pt.read_accum()
            assert np.array_equal(cnt, pcnt) and np.array_equal(s4, p4), (cfg, stage)
            assert pt.stats().connect_rays == pst.connect_rays


def test_light_geometry_with_an_inner_root_ignores_its_meaningless_range(pt):
    # start/end of an inner node mean nothing (src/bvh.rs:206): a light whose BLAS
    # root is an inner node is sampled over all triangles of its geometry, whatever
    # a client leaves in those two fields
    s = host.Scene()
    cube, rect = s.load_cube(), s.load_rect()
    lamp = s.add_material(host.make_material(colour=(1, 1, 1, 1), emissive=(15, 15, 15, 1)))
    grey = s.add_material(host.make_material(colour=(0.7, 0.7, 0.7, 1.0)))
    s.spawn(host.make_transform(scale=(0.4, 0.1, 0.4), translation=(0.0, 1.2, 3.0)), lamp, cube)
    s.spawn(host.make_transform(scale=(3, 3, 3), rotation=(np.pi / 2, 0, 0), translation=(0.0, -1.0, 3.0)), grey, rect)
    s.spawn(host.make_transform(scale=(0.5, 0.5, 0.5), translation=(0.3, -0.4, 3.0)), grey, cube)
    arrays = s.bind()
    s.close()
    assert arrays["light_sources"].tolist() == [0]
    g = arrays["geometry_offsets"][arrays["instances"]["geometry"][0]]
    root = int(g["blas_node"])
    assert arrays["blas_nodes"]["is_leaf"][root] == 0  # twelve triangles: the root was split
    W, H = 96, 64
    src = host.make_sample_sources(W, H, 9)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=4, seed=9, shade_mode=1, nee=1)
    oprm = po.default_params(max_bounces=4, seed=9, shade_mode=1, nee=1)
    w4, wcnt, wst = po.Scene(arrays).render(cam, src, W, H, oprm, 0, 2)
    assert wst.connect_rays > 1000
    for start, end in ((0, 0), (0xFFFFFFF0, 7)):
        odd = {k: v.copy() for k, v in arrays.items()}
        odd["blas_nodes"]["start"][root] = start
        odd["blas_nodes"]["end"][root] = end
        pt.upload_scene(odd)
        pt.create_state(W, H, 3000, src, prm)
        pt.set_camera(cam)
        pt.render_spp(2)
        s4, cnt = pt.read_accum()
        assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
        o4, ocnt, _ = po.Scene(odd).render(cam, src, W, H, oprm, 0, 2)
        assert np.array_equal(o4, w4) and np.array_equal(ocnt, wcnt)


def test_shadow_ray_sets_any_hit_ids(pt, config_scenes):
    # the connect (shadow) rays of configuration 5's scene at bounces 1 and 2, as
    # the oracle's per-material path generates them, traced as any-hit queries:
    # occlusion flag, first accepted triangle, t bits and visit counts
    arrays, info = config_scenes(5, dragon_n_u=96, dragon_n_v=24)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = 160, 90
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    oprm = po.default_params(max_bounces=8, seed=info.seed, shade_mode=1, nee=1)
    total = 0
    for bounce in (1, 2):
        rays, tmax = sc.connect_rays(cam, src, W, H, oprm, 0, bounce)
        assert len(rays) > 500
        total += len(rays)
        want, wc = sc.trace_any(rays, tmax, want_counts=True)
        got, gc = pt.trace_any(rays, tmax, want_counts=True)
        assert hits_equal(got, want)
        assert gc.as_dict() == wc.as_dict()
        occluded = want["instance_id"] != ffi.NO_HIT
        assert 0 < occluded.sum() < len(rays)  # both outcomes occur
    assert total > 2000


def test_config1_full_render_and_pixels(pt, config_scenes):
    # configuration 1 in full: suzanne + light, 256x256, 1 spp, 4 bounces
    arrays, info = config_scenes(1)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = info.width, info.height
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed)
    pt.create_state(W, H, W * H, src, prm)
    pt.set_camera(cam)
    pt.set_counting(True)
    pt.stats_reset()
    pt.render_spp(info.spp)
    pt.set_counting(False)
    s4, cnt = pt.read_accum()
    w4, wcnt, wst = sc.render(cam, src, W, H, oprm, 0, info.spp)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
    assert np.array_equal(pt.read_output(), po.resolve(w4, wcnt, src, W, H, oprm))
    st = pt.stats()
    assert st.extend_counts.as_dict() == wst.extend_counts.as_dict()
    # a second call continues with the next sample numbers
    pt.render_spp(2)
    s4b, cntb = pt.read_accum()
    sc.render(cam, src, W, H, oprm, info.spp, 2, sum4=w4, count=wcnt)
    assert np.array_equal(s4b, w4) and np.array_equal(cntb, wcnt)


def test_render_with_cooperative_leaves_and_many_bounces(pt, small_knot_scene):
    arrays = small_knot_scene
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = 160, 90
    src = host.make_sample_sources(W, H, 5)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=8, seed=77)
    oprm = po.default_params(max_bounces=8, seed=77)
    pt.create_state(W, H, 4096, src, prm)
    pt.set_camera(cam)
    pt.render_spp(2)
    s4, cnt = pt.read_accum()
    w4, wcnt, _ = sc.render(cam, src, W, H, oprm, 0, 2)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)


def test_bounce_rays_dumped_from_the_device_match_the_oracle(pt, config_scenes):
    arrays, info = config_scenes(2)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    W, H = 96, 54
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=8, seed=info.seed)
    oprm = po.default_params(max_bounces=8, seed=info.seed)
    pt.create_state(W, H, W * H, src, prm)  # one slot per source
    pt.set_camera(cam)
    pt.step(2)
    pt.synchronize()
    got = pt.dump_extension_rays(W * H)
    # After two frames the extension queue holds what the second shade launch
    # pushed: the bounce-2 rays of sample 0 of every source, plus the bounce-1
    # rays of the samples respawned in frame 1 — sample 1 of sources 0..n1-1,
    # where n1 slots were freed by sample-0 paths that missed everything.
    b2 = sc.bounce_rays(cam, src, W, H, oprm, 0, 2)
    b2 = b2[b2["last_inst"] != ffi.NO_HIT]
    b1_k0 = sc.bounce_rays(cam, src, W, H, oprm, 0, 1)
    n1 = int((b1_k0["last_inst"] == ffi.NO_HIT).sum())
    b1 = sc.bounce_rays(cam, src[:n1].copy(), W, H, oprm, 1, 1)
    b1 = b1[b1["last_inst"] != ffi.NO_HIT]
    want = np.concatenate([b2, b1])
    assert len(got) == len(want) > 100

    def key(r):
        return np.sort(r.view("<u4").reshape(-1, 8).view([("k", "<u4", 8)]).reshape(-1), order="k")

    assert np.array_equal(key(got), key(want))  # queue order is free, the set is not


@pytest.mark.parametrize("cfg,W,H,slots,frames", [(1, 64, 64, 64 * 64, 9), (2, 64, 36, 1000, 23)])
def test_step_frames_accumulate_what_the_wavefront_simulation_does(pt, config_scenes, cfg, W, H, slots, frames):
    # mirt_step is the call a per-frame host makes (pathtracer_phase_execute,
    # src/pathtracer_manager.rs:74-124): F frames of it — the three-kernel frame, samples
    # regenerating without a budget — against F frames of the oracle's queue simulation:
    # sums, counts and the ray / shade / sample statistics, in both shading modes, with as
    # many slots as pixels (the reference's shape) and with far fewer
    arrays, info = config_scenes(cfg)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    for mode in (0, 1):
        prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode)
        oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode)
        pt.create_state(W, H, slots, src, prm)
        pt.set_camera(cam)
        pt.stats_reset()
        pt.step(frames - 3)
        pt.step(3)  # frames of separate calls continue the same wavefront
        pt.synchronize()
        s4, cnt = pt.read_accum()
        w4, wcnt, wst = sc.step_wavefront(cam, src, W, H, oprm, 0, frames, slots)
        assert wst.finished_samples > slots  # slots were recycled
        assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
        st = pt.stats()
        assert (st.extend_rays, st.connect_rays, st.shade_events, st.finished_samples) == (
            wst.extend_rays, wst.connect_rays, wst.shade_events, wst.finished_samples)
        assert st.frames == frames and st.overflowed_sums == 0


def test_errors_are_reported_not_thrown(pt, config_scenes):
    arrays, _ = config_scenes(1)
    bad = {k: v.copy() for k, v in arrays.items()}
    bad["blas_nodes"]["right"][3] = 10 ** 6
    with pytest.raises(pathtracer.MirtError, match="skip link"):
        pt.upload_scene(bad)
    bad = {k: v.copy() for k, v in arrays.items()}
    bad["blas_nodes"]["right"][1] = 1  # a skip link onto itself: the walk would never end
    with pytest.raises(pathtracer.MirtError, match="cycle"):
        pt.upload_scene(bad)
    bad = {k: v.copy() for k, v in arrays.items()}
    bad["instances"]["material"][0] = 7
    with pytest.raises(pathtracer.MirtError):
        pt.upload_scene(bad)
    # two leaves over the same triangles (they would overwrite each other's per-leaf words)
    bad = {k: v.copy() for k, v in arrays.items()}
    leaves = np.flatnonzero(bad["blas_nodes"]["is_leaf"] == 1)
    bad["blas_nodes"]["start"][leaves[1]] = bad["blas_nodes"]["start"][leaves[0]]
    bad["blas_nodes"]["end"][leaves[1]] = bad["blas_nodes"]["end"][leaves[0]]
    with pytest.raises(pathtracer.MirtError, match="overlap"):
        pt.upload_scene(bad)
    fresh = pathtracer.Pathtracer(0)
    with pytest.raises(pathtracer.MirtError, match="no scene"):
        fresh.trace_closest(random_rays(4, 1))
    fresh.upload_scene(arrays)
    with pytest.raises(pathtracer.MirtError, match="state"):
        fresh.render_spp(1)
    # per-material shading on a scene uploaded without its per-material tables
    bare = {k: v for k, v in arrays.items() if k not in ("classic_refs", "lambertian", "metallic", "dielectric", "emissive")}
    fresh.upload_scene(bare)
    ok_src = host.make_sample_sources(8, 8, 1)
    fresh.create_state(8, 8, 64, ok_src, pathtracer.default_params(max_bounces=2, shade_mode=1))
    fresh.set_camera(host.default_camera(8, 8))
    with pytest.raises(pathtracer.MirtError, match="per-material tables"):
        fresh.render_spp(1)
    fresh.upload_scene(arrays)
    src = host.make_sample_sources(8, 8, 1)
    src["out_pos"][0] = (99, 0)
    with pytest.raises(pathtracer.MirtError, match="outside"):
        fresh.create_state(8, 8, 64, src, pathtracer.default_params())
    fresh.close()
    L = ffi.device_lib()
    assert L.mirt_create(10 ** 6, ffi.C.byref(ffi.C.c_void_p())) != 0
    assert L.mirt_scene_upload(None, None) != 0


def test_full_size_properties_config3_1080p(pt, config_scenes):
    # the benchmark workload at full size: properties that need no oracle run
    arrays, info = config_scenes(3)
    pt.upload_scene(arrays)
    W, H = info.width, info.height
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)
    pt.create_state(W, H, W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(2)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == 2) and st.finished_samples == 2 * W * H
    assert st.extend_rays >= st.finished_samples and st.shade_events <= st.extend_rays
    # every path is either terminated by the sky or runs out of bounces
    assert st.extend_rays - st.shade_events <= st.finished_samples
    # sample ranges add up: [0,1) + [1,2) == [0,2), with a different slot count
    pt.create_state(W, H, 700001, src, prm)
    pt.set_camera(cam)
    pt.render_spp(1)
    pt.render_spp(1)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
    # a slice of it against the oracle: the first 4096 sources' sums
    sc = po.Scene(arrays)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed)
    w4, wc, _ = sc.render(cam, src[:4096].copy(), W, H, oprm, 0, 2)
    assert np.array_equal(a4[: 4 * 4096], w4) and np.array_equal(ac[:4096], wc)
    # sky-only furnace at full size
    cam2 = cam.copy()
    cam2["forward"][0] = (0.0, 0.0, -1.0)
    cam2["changed"] = 1
    pt.set_camera(cam2)
    pt.render_spp(1)
    f4, fc = pt.read_accum()
    assert np.all(fc == 1) and np.all(f4.reshape(-1, 4)[:, :3] == 10000)


def test_config3_at_its_full_sample_count(pt, config_scenes):
    # the headline configuration at its own size: 1920x1080, 256 samples per pixel, 8 bounces.
    # One call of 256 spp equals four calls of 64 with another number of slots, every source
    # gets exactly 256 samples, and no fixed-point sum comes near the 32-bit range
    arrays, info = config_scenes(3)
    assert info.spp == 256
    pt.upload_scene(arrays)
    W, H = info.width, info.height
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)
    pt.create_state(W, H, 16 * W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(256)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == 256) and st.finished_samples == 256 * W * H == st.spawned_samples
    assert int(a4.max()) < 2 ** 31
    pt.create_state(W, H, 24 * W * H + 17, src, prm)
    pt.set_camera(cam)
    for _ in range(4):
        pt.render_spp(64)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)


def test_full_size_properties_per_material_path_1080p(pt, config_scenes):
    # configuration 5's scene with the full-size dragon, per-material shading with
    # light sampling, at 1080p: properties that need no full oracle run
    arrays, info = config_scenes(5)
    assert info.shade_mode == 1 and info.nee == 1
    pt.upload_scene(arrays)
    W, H = 1920, 1080
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=1, nee=1)
    pt.create_state(W, H, 3 * W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(2)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == 2) and st.finished_samples == 2 * W * H == st.spawned_samples
    # one connect ray at most per hit that is shaded, none without a hit
    assert 0 < st.connect_rays <= st.shade_events <= st.extend_rays
    # sample ranges add up and the result does not depend on the number of slots
    pt.create_state(W, H, 500003, src, prm)
    pt.set_camera(cam)
    pt.render_spp(1)
    pt.render_spp(1)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
    # a slice of it against the oracle: the first 2048 sources' sums
    sc = po.Scene(arrays)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=1, nee=1)
    w4, wc, _ = sc.render(cam, src[:2048].copy(), W, H, oprm, 0, 2)
    assert np.array_equal(a4[: 4 * 2048], w4) and np.array_equal(ac[:2048], wc)


@pytest.mark.parametrize("cfg", [3, 5])
def test_launches_that_stage_their_rays_match_the_oracle_slice(pt, config_scenes, cfg):
    # The walk comes by its rays in two ways: launches of at least 16 x 128 entries per resident wave (8.4 M on
    # 256 CUs) fetch them 64 at a time ahead of need, smaller ones when a lane asks.  With twelve slots per pixel
    # at 1080p the first launches of both walks (closest hit, and shadow rays for configuration 5) are of the first
    # kind; the same samples through one slot per pixel are of the second; both must be the oracle's.
    arrays, info = config_scenes(cfg)
    pt.upload_scene(arrays)
    W, H, spp = 1920, 1080, 12
    assert spp * W * H >= 2 * 16 * 128 * 16 * 256
    kw = dict(max_bounces=info.max_bounces, seed=info.seed, shade_mode=info.shade_mode, nee=info.nee)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(**kw)
    pt.create_state(W, H, spp * W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(spp)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == spp) and st.finished_samples == spp * W * H == st.spawned_samples
    if info.nee:
        assert st.connect_rays > 8400000
    pt.create_state(W, H, W * H, src, prm)
    pt.set_camera(cam)
    pt.render_spp(spp)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
    sc = po.Scene(arrays)
    w4, wc, _ = sc.render(cam, src[:2048].copy(), W, H, po.default_params(**kw), 0, spp)
    assert np.array_equal(a4[: 4 * 2048], w4) and np.array_equal(ac[:2048], wc)


def test_full_size_properties_config4_thousand_instances_1080p(pt, config_scenes):
    # configuration 4 at full size (1000 instances of the full-size dragon stand-in, teapot,
    # suzanne and cube under one TLAS): ray-set parity on a sample, a slice of the render
    # against the oracle, and the size-independent properties
    arrays, info = config_scenes(4)
    assert len(arrays["instances"]) == 1000
    pt.upload_scene(arrays)
    rays = random_rays(20000, 44, centre=(0.0, 0.0, 14.0), spread=9.0)
    sc = po.Scene(arrays)
    got, gc = pt.trace_closest(rays, want_counts=True)
    want, wc = sc.trace_closest(rays, want_counts=True)
    assert hits_equal(got, want)
    assert (gc.tlas_nodes, gc.instance_visits, gc.blas_nodes, gc.triangles) == \
           (wc.tlas_nodes, wc.instance_visits, wc.blas_nodes, wc.triangles)
    W, H = info.width, info.height
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed)
    pt.create_state(W, H, W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(2)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == 2) and st.finished_samples == 2 * W * H == st.spawned_samples
    assert st.extend_rays - st.shade_events <= st.finished_samples
    pt.create_state(W, H, 300007, src, prm)
    pt.set_camera(cam)
    pt.render_spp(1)
    pt.render_spp(1)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed)
    w4, wcnt, _ = sc.render(cam, src[:1024].copy(), W, H, oprm, 0, 2)
    assert np.array_equal(a4[: 4 * 1024], w4) and np.array_equal(ac[:1024], wcnt)


def test_full_size_properties_config5_4k(pt, config_scenes):
    # configuration 5 at its own size, 3840x2160 (8.3 M sources), per-material shading with
    # light sampling: every source gets its sample, slot count does not matter, a slice
    # equals the oracle
    arrays, info = config_scenes(5)
    pt.upload_scene(arrays)
    W, H = info.width, info.height
    assert (W, H) == (3840, 2160)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=1, nee=1)
    pt.create_state(W, H, W * H, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(1)
    a4, ac = pt.read_accum()
    st = pt.stats()
    assert np.all(ac == 1) and st.finished_samples == W * H == st.spawned_samples
    assert 0 < st.connect_rays <= st.shade_events <= st.extend_rays
    pt.create_state(W, H, 1000003, src, prm)
    pt.set_camera(cam)
    pt.render_spp(1)
    b4, bc = pt.read_accum()
    assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
    sc = po.Scene(arrays)
    oprm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=1, nee=1)
    w4, wc, _ = sc.render(cam, src[:1024].copy(), W, H, oprm, 0, 1)
    assert np.array_equal(a4[: 4 * 1024], w4) and np.array_equal(ac[:1024], wc)


def test_plain_c_driver_renders_the_same_sums(pt, config_scenes, tmp_path):
    # examples/render_c.c (built by `make examples`) goes through both C ABIs from
    # C: its checksum of the radiance sums must equal this process's render
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "lib", "render_c")
    assert os.path.exists(exe), "run `make examples`"
    assets = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    W, H, spp = 96, 64, 3
    out = subprocess.run([exe, assets, "1", str(spp), str(tmp_path / "c.ppm"), str(W), str(H)], capture_output=True,
                         text=True, timeout=100)
    assert out.returncode == 0, out.stderr
    words = out.stdout.split()
    arrays, info = config_scenes(1)
    pt.upload_scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    pt.create_state(W, H, W * H, src, pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed))
    pt.set_camera(host.default_camera(W, H))
    pt.render_spp(spp)
    s4, cnt = pt.read_accum()
    want = int(s4.reshape(-1, 4)[:, :3].astype(np.uint64).sum())
    assert int(words[1]) == want and int(words[3]) == int(cnt.sum()) == W * H * spp
    ppm = open(tmp_path / "c.ppm", "rb").read()
    assert ppm.startswith(b"P6\n96 64\n255\n") and len(ppm) == len(b"P6\n96 64\n255\n") + 3 * W * H
    # the same render through a group of two contexts (mirt_group_*): same sums, same image
    out2 = subprocess.run([exe, assets, "1", str(spp), str(tmp_path / "g.ppm"), str(W), str(H), "2"], capture_output=True,
                          text=True, timeout=100)
    assert out2.returncode == 0, out2.stderr
    assert out2.stdout.split() == words
    assert open(tmp_path / "g.ppm", "rb").read() == ppm
    # and with the BLASes built on the device through the host model's builder callback
    out3 = subprocess.run([exe, assets, "1", str(spp), str(tmp_path / "d.ppm"), str(W), str(H)], capture_output=True,
                          text=True, timeout=100, env=dict(os.environ, MIRT_DEVICE_BUILD="1"))
    assert out3.returncode == 0, out3.stderr
    assert out3.stdout.split() == words
    assert open(tmp_path / "d.ppm", "rb").read() == ppm


def _three_suzannes(moved):
    # the same two meshes, instances placed differently (and one more of them when `moved`)
    s = host.Scene()
    suz = s.load_model_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "suzanne.mirtmesh"))
    rect = s.load_rect()
    grey = s.add_material(host.make_material(colour=(0.7, 0.7, 0.7, 1.0)))
    lamp = s.add_material(host.make_material(emissive=(12.0, 12.0, 12.0, 0.0)))
    places = [((-1.2, 0.0, 3.0), (0.0, 0.3, 0.0)), ((1.2, 0.0, 3.5), (0.0, -0.4, 0.0)), ((0.0, 1.0, 4.5), (0.2, 0.0, 0.0))]
    if moved:
        places = [((-0.4, 0.3, 2.6), (0.5, 1.1, 0.0)), ((1.5, -0.5, 4.0), (0.0, 2.0, 0.3)), ((0.2, 0.8, 5.5), (0.0, 0.0, 1.0)),
                  ((0.0, -1.0, 3.2), (1.0, 0.0, 0.0))]
    for pos, rot in places:
        s.spawn(host.make_transform(scale=(0.8, 0.8, 0.8), rotation=rot, translation=pos), grey, suz)
    s.spawn(host.make_transform(scale=(2.0, 2.0, 1.0), rotation=(1.5707964, 0.0, 0.0), translation=(0.0, 2.5, 3.5)), lamp, rect)
    arrays = s.bind()
    s.close()
    return arrays


def test_instance_update_keeps_geometry_and_matches_a_fresh_upload(pt):
    # mirt_scene_update_instances: transforms / instance count / TLAS change, the
    # BLAS nodes and triangles stay in HBM; results equal a fresh upload and the oracle
    a, b = _three_suzannes(False), _three_suzannes(True)
    for k in ("geometry_offsets", "vertices", "indices", "blas_nodes"):
        assert np.array_equal(a[k], b[k])
    assert len(b["instances"]) == len(a["instances"]) + 1
    rays = random_rays(50021, 5, centre=(0.0, 0.0, 3.5), spread=2.5)
    pt.upload_scene(a)
    assert hits_equal(pt.trace_closest(rays), po.Scene(a).trace_closest(rays))
    pt.update_instances(b)
    sc = po.Scene(b)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want) and gc.as_dict() == wc.as_dict()
    fresh = pathtracer.Pathtracer(0)
    fresh.upload_scene(b)
    assert hits_equal(fresh.trace_closest(rays), want)
    fresh.close()
    # and a render on the updated scene
    W, H = 64, 48
    src = host.make_sample_sources(W, H, 9)
    cam = host.default_camera(W, H)
    pt.create_state(W, H, 2000, src, pathtracer.default_params(max_bounces=5, seed=9))
    pt.set_camera(cam)
    pt.render_spp(2)
    s4, cnt = pt.read_accum()
    w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(max_bounces=5, seed=9), 0, 2)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
    # back again, then a broken update is refused and leaves an error, not a crash
    pt.update_instances(a)
    assert hits_equal(pt.trace_closest(rays), po.Scene(a).trace_closest(rays))
    bad = {k: v.copy() for k, v in b.items()}
    bad["tlas_to_instances"][0] = 99
    with pytest.raises(pathtracer.MirtError, match="tlas_to_instances"):
        pt.update_instances(bad)


def test_degenerate_trees_single_instance_single_leaf(pt):
    # one rect: the TLAS root is a leaf and so is the BLAS root (every walk wraps at once)
    s = host.Scene()
    rect = s.load_rect()
    m = s.add_material(host.make_material(emissive=(3.0, 3.0, 3.0, 0.0)))
    s.spawn(host.make_transform(scale=(2.0, 2.0, 1.0), translation=(0.0, 0.0, 3.0)), m, rect)
    arrays = s.bind()
    s.close()
    assert len(arrays["tlas_nodes"]) == 1 and arrays["tlas_nodes"]["is_leaf"][0] == 1
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    rays = random_rays(20011, 8, centre=(0.0, 0.0, 3.0), spread=1.5)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want) and gc.as_dict() == wc.as_dict()
    assert 100 < (want["instance_id"] != ffi.NO_HIT).sum() < len(rays)
    tmax = np.full(len(rays), 2.5, "<f4")
    assert hits_equal(pt.trace_any(rays, tmax), sc.trace_any(rays, tmax))
    W, H = 48, 32
    src = host.make_sample_sources(W, H, 4)
    cam = host.default_camera(W, H)
    pt.create_state(W, H, 300, src, pathtracer.default_params(max_bounces=3, seed=4))
    pt.set_camera(cam)
    pt.render_spp(2)
    s4, cnt = pt.read_accum()
    w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(max_bounces=3, seed=4), 0, 2)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)


@pytest.mark.gpu
def test_bench_collective_path_with_one_rank():
    """bench.py's multi-GPU branch (RCCL process group, device-to-device accumulator
    all-reduce, MAX / SUM of the per-rank figures) rehearsed with a single rank."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIRT_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--config", "1", "--spp-per-step", "4", "--paths-per-pixel", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["rays"] > 0
    assert d["reduce_ms"] > 0.0  # the collective ran
    assert d["roofline"]["achieved"] > 0 and d["roofline"]["launches"] > 0
    assert d["config"]["parallelism"] == "spp split x1"


def _bench_line(argv, env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


SMALL_BENCH = ["--steps", "2", "--warmup", "1", "--config", "1", "--spp-per-step", "4", "--paths-per-pixel", "2",
               "--no-cpu-baseline"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_reduces_across_them():
    """`python bench.py --gpus 2` as the driver runs it, with no launcher around it: two rank processes, an
    all-reduce of their accumulators inside the timed region, ONE line from rank 0 that counts both ranks' rays.
    Both ranks share this box's one GPU here, so the process group is gloo (RCCL refuses two ranks on a device)."""
    one = _bench_line(["--gpus", "1"] + SMALL_BENCH, {})
    two = _bench_line(["--gpus", "2"] + SMALL_BENCH, {"MIRT_BENCH_DEVICES": "0,0", "MIRT_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["reduce_ms"] > 0.0
    assert "one process per GPU" in two["config"]["parallelism"]
    # weak scaling: every rank renders the samples one GPU renders, on its own sample numbers
    assert 1.9 * one["rays"] < two["rays"] < 2.1 * one["rays"]
    assert 0 < two["per_rank_mrays_per_s"]["min"] <= two["per_rank_mrays_per_s"]["max"]
    assert two["cpu_baseline"] is None and two["roofline"]["launches"] > 0


@pytest.mark.gpu
def test_bench_group_mode_drives_its_members_from_one_process():
    grp = _bench_line(["--gpus", "2", "--group"] + SMALL_BENCH, {"MIRT_BENCH_DEVICES": "0,0"})
    one = _bench_line(["--gpus", "1"] + SMALL_BENCH, {})
    assert grp["n_gpus"] == 2 and "mirt_group" in grp["config"]["parallelism"] and grp["reduce_ms"] > 0.0
    assert 1.9 * one["rays"] < grp["rays"] < 2.1 * one["rays"]


def _comb_scene(leaf_sizes, seed):
    """One instance of a triangle soup under a hand-built comb BVH whose leaves have the
    given sizes: root -> (leaf 0, inner) -> (leaf 1, inner) ... -> (leaf k-2, leaf k-1)."""
    s = host.Scene()
    rect = s.load_rect()
    m = s.add_material(host.make_material(colour=(0.7, 0.7, 0.7, 1.0)))
    s.spawn(host.make_transform(), m, rect)
    arrays = dict(s.bind())
    s.close()
    rng = np.random.default_rng(seed)
    n = int(sum(leaf_sizes))
    c = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.0, 4.0, n)], 1)
    corners = c[:, None, :] + rng.uniform(-0.25, 0.25, (n, 3, 3))
    verts = np.zeros(3 * n, arrays["vertices"].dtype)
    verts["position"][:, :3] = corners.reshape(-1, 3).astype("<f4")
    verts["position"][:, 3] = 1.0
    verts["normal"][:, 2] = -1.0
    idx = np.zeros(n, arrays["indices"].dtype)
    idx["v"][:, :3] = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    k = len(leaf_sizes)
    nodes = np.zeros(2 * k - 1, arrays["blas_nodes"].dtype)
    starts = np.concatenate([[0], np.cumsum(leaf_sizes)]).astype(np.uint32)
    pos = verts["position"][:, :3].reshape(n, 3, 3)

    def leaf(at, i, skip):
        nodes[at]["is_leaf"] = 1
        nodes[at]["start"], nodes[at]["end"] = starts[i], starts[i + 1]
        nodes[at]["right"] = skip
        p = pos[starts[i]:starts[i + 1]].reshape(-1, 3)
        nodes[at]["lb"][:3], nodes[at]["ub"][:3] = p.min(0), p.max(0)

    nodes[0]["left"], nodes[0]["right"] = 1, 0
    for i in range(k - 1):
        last = i == k - 2
        leaf(2 * i + 1, i, 2 * i + 2)
        if last:
            leaf(2 * i + 2, i + 1, 0)
        else:
            nodes[2 * i + 2]["left"], nodes[2 * i + 2]["right"] = 2 * i + 3, 0
    for at in range(2 * k - 2, -1, -1):  # inner boxes, children first
        if nodes[at]["is_leaf"] != 1:
            l = int(nodes[at]["left"])
            nodes[at]["lb"][:3] = np.minimum(nodes[l]["lb"][:3], nodes[l + 1]["lb"][:3])
            nodes[at]["ub"][:3] = np.maximum(nodes[l]["ub"][:3], nodes[l + 1]["ub"][:3])
            nodes[at]["start"], nodes[at]["end"] = nodes[l]["start"], nodes[l + 1]["end"]
    arrays["vertices"], arrays["indices"], arrays["blas_nodes"] = verts, idx, nodes
    tl = arrays["tlas_nodes"].copy()
    tl["lb"][0][:3], tl["ub"][0][:3] = nodes[0]["lb"][:3], nodes[0]["ub"][:3]
    arrays["tlas_nodes"] = tl
    return arrays


def test_whole_wave_leaves_of_every_block_shape(pt):
    """Leaves around every boundary of the whole-wave scan: its threshold (23 / 24), one
    64-triangle block and its neighbours (63 / 64 / 65), several blocks, a ragged tail, and
    the sizes whose count is stored out of line (254 / 255 / 256), mixed with small leaves."""
    sizes = [24, 63, 1, 64, 65, 4, 128, 23, 254, 255, 256, 300, 2, 777]
    arrays = _comb_scene(sizes, 12)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    rays = random_rays(30011, 13, centre=(0.0, 0.0, 3.0), spread=2.0)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want) and gc.as_dict() == wc.as_dict()
    hit = want["instance_id"] != ffi.NO_HIT
    assert 3000 < hit.sum() < len(rays)
    # every scanned leaf is somebody's closest hit, so every block shape has produced a winner
    starts = np.concatenate([[0], np.cumsum(sizes)])
    leaf_of = np.searchsorted(starts, want["triangle_id"][hit], side="right") - 1
    assert set(leaf_of.tolist()) >= {i for i, n in enumerate(sizes) if n >= 23}
    tmax = np.full(len(rays), 2.0, "<f4")
    want_any, wca = sc.trace_any(rays, tmax, True)
    got_any, gca = pt.trace_any(rays, tmax, True)
    assert hits_equal(got_any, want_any) and gca.as_dict() == wca.as_dict()
    # bounce rays leave from triangles inside those leaves: the excluded triangle sits in a scanned block
    W, H = 64, 48
    src = host.make_sample_sources(W, H, 6)
    cam = host.default_camera(W, H)
    pt.create_state(W, H, 1000, src, pathtracer.default_params(max_bounces=4, seed=6))
    pt.set_camera(cam)
    pt.render_spp(3)
    s4, cnt = pt.read_accum()
    w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(max_bounces=4, seed=6), 0, 3)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)


def test_small_leaves_of_every_size_tested_four_lanes_to_a_leaf(pt):
    """Leaves below the whole-wave threshold are tested at the service point, four lanes to a leaf
    and sixteen leaves per pass: every size 1..23 (one to six passes of a lane over its leaf), far
    more leaves than one pass takes, closest and any hit with visit counts, and bounce rays whose
    excluded triangle sits inside such a leaf."""
    rng = np.random.default_rng(23)
    sizes = [int(n) for n in rng.permutation(np.repeat(np.arange(1, 24), 3))]
    arrays = _comb_scene(sizes, 12)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    rays = random_rays(40009, 29, centre=(0.0, 0.0, 3.0), spread=2.0)
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want) and gc.as_dict() == wc.as_dict()
    hit = want["instance_id"] != ffi.NO_HIT
    starts = np.concatenate([[0], np.cumsum(sizes)])
    leaf_of = np.searchsorted(starts, want["triangle_id"][hit], side="right") - 1
    assert {sizes[i] for i in set(leaf_of.tolist())} == set(range(1, 24))
    tmax = np.full(len(rays), 2.0, "<f4")
    want_any, wca = sc.trace_any(rays, tmax, True)
    got_any, gca = pt.trace_any(rays, tmax, True)
    assert hits_equal(got_any, want_any) and gca.as_dict() == wca.as_dict()
    W, H = 64, 48
    src = host.make_sample_sources(W, H, 8)
    cam = host.default_camera(W, H)
    pt.create_state(W, H, 1000, src, pathtracer.default_params(max_bounces=4, seed=8))
    pt.set_camera(cam)
    pt.render_spp(3)
    s4, cnt = pt.read_accum()
    w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(max_bounces=4, seed=8), 0, 3)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)


def test_group_of_contexts_splits_samples_and_reduces_bit_exactly(config_scenes):
    """mirt_group_*: three contexts (all on device 0 here; one per GPU on a node), samples
    split 2 + 2 + 1 and then 2 + 1 + 1, integer reduction into member 0 — the same sums as
    one context and as the oracle rendering 9 samples."""
    arrays, info = config_scenes(1)
    W, H = 96, 64
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=4, seed=info.seed)
    with pathtracer.PathtracerGroup([0, 0, 0]) as grp:
        assert len(grp) == 3
        grp.upload_scene(arrays)
        grp.create_state(W, H, 3000, src, prm)
        grp.set_camera(cam)
        grp.render_spp(5)
        parts = [m.read_accum() for m in grp.members]
        assert [int(c.sum()) for _, c in parts] == [2 * W * H, 2 * W * H, W * H]
        grp.reduce()
        s5, c5 = grp.read_accum()
        assert np.array_equal(s5, parts[0][0] + parts[1][0] + parts[2][0]) and np.array_equal(c5, np.full(W * H, 5, "<u4"))
        for m in grp.members[1:]:  # cleared, so that a second reduction adds nothing twice
            s, c = m.read_accum()
            assert not s.any() and not c.any()
        grp.render_spp(4)
        grp.reduce()
        s9, c9 = grp.read_accum()
        pixels = grp.read_output()
    with pathtracer.Pathtracer(0) as one:
        one.upload_scene(arrays)
        one.create_state(W, H, 3000, src, prm)
        one.set_camera(cam)
        one.render_spp(9)
        w4, wc = one.read_accum()
        wpix = one.read_output()
    assert np.array_equal(s9, w4) and np.array_equal(c9, wc) and np.array_equal(pixels, wpix)
    o4, oc, _ = po.Scene(arrays).render(cam, src, W, H, po.default_params(max_bounces=4, seed=info.seed), 0, 9)
    assert np.array_equal(s9, o4) and np.array_equal(c9, oc)
    # errors carry the member and its message
    with pathtracer.PathtracerGroup([0]) as g1:
        with pytest.raises(pathtracer.MirtError, match="member 0"):
            g1.render_spp(1)
    with pytest.raises(pathtracer.MirtError):
        pathtracer.PathtracerGroup([9999])


def test_axis_aligned_rays_from_box_planes_nan_semantics(pt):
    """Rays with zero direction components whose origins lie exactly on box planes: the slab
    test then multiplies 0 by infinity, and what min / max do with the NaN decides the walk
    (rayBoxIntersect, shaders/ray_extend.slang:94-111; here: the NaN is ignored).  Hit ids,
    t bits and visit counts must still be the oracle's."""
    arrays = _comb_scene([3, 30, 4, 70, 2, 4, 1, 90], 21)
    sc = po.Scene(arrays)
    pt.upload_scene(arrays)
    nodes = arrays["blas_nodes"]
    rng = np.random.default_rng(22)
    n = 24000
    pick = rng.integers(0, len(nodes), n)
    corner = rng.integers(0, 2, (n, 3)).astype(bool)
    pos = np.where(corner, nodes["ub"][pick][:, :3], nodes["lb"][pick][:, :3]).astype("<f4")
    # a third of the origins are pushed off the planes along one axis, so that rays also start outside
    off = rng.integers(0, 3, n)
    pos[np.arange(n), off] += np.where(rng.random(n) < 0.33, rng.choice(np.float32([-3.0, 3.0]), n), np.float32(0.0))
    d = np.zeros((n, 3), "<f4")
    kind = rng.integers(0, 3, n)
    axis = rng.integers(0, 3, n)
    sign = rng.choice(np.float32([-1.0, 1.0]), n)
    d[np.arange(n), axis] = sign                                    # one non-zero component
    two = kind >= 1                                                 # two non-zero components
    d[np.arange(n)[two], (axis[two] + 1) % 3] = rng.normal(size=two.sum()).astype("<f4")
    neg0 = kind == 2                                                # the zero component is -0.0: 1/dir = -inf
    d[np.arange(n)[neg0], (axis[neg0] + 2) % 3] = np.float32(-0.0)
    rays = np.zeros(n, ffi.Ray)
    rays["pos"], rays["dir"] = pos, d
    rays["last_inst"] = ffi.NO_PRIM
    rays["last_prim"] = ffi.NO_PRIM
    want, wc = sc.trace_closest(rays, True)
    got, gc = pt.trace_closest(rays, True)
    assert hits_equal(got, want) and gc.as_dict() == wc.as_dict()
    hit = want["instance_id"] != ffi.NO_HIT
    assert 500 < hit.sum() < n
    tmax = np.full(n, 1.5, "<f4")
    want_any, wca = sc.trace_any(rays, tmax, True)
    got_any, gca = pt.trace_any(rays, tmax, True)
    assert hits_equal(got_any, want_any) and gca.as_dict() == wca.as_dict()


def test_group_on_two_distinct_devices_when_the_box_has_them(config_scenes):
    """The peer copies of mirt_group_reduce between two different GPUs (skipped on a one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    arrays, info = config_scenes(1)
    W, H = 96, 64
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=4, seed=info.seed)
    with pathtracer.PathtracerGroup([0, 1]) as grp:
        grp.upload_scene(arrays)
        grp.create_state(W, H, 3000, src, prm)
        grp.set_camera(cam)
        grp.render_spp(5)
        grp.reduce()
        s5, c5 = grp.read_accum()
    o4, oc, _ = po.Scene(arrays).render(cam, src, W, H, po.default_params(max_bounces=4, seed=info.seed), 0, 5)
    assert np.array_equal(s5, o4) and np.array_equal(c5, oc)


def _loaded_hip_runtime():
    """The HIP runtime the device library has already loaded into this process (a second copy, such as the one
    torch ships, finds no device once the first holds it)."""
    with open("/proc/self/maps") as f:
        paths = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    assert paths, "the device library has not loaded a HIP runtime"
    hip = C.CDLL(sorted(paths)[0])
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    return hip


def test_radiance_sums_saturate_and_report_instead_of_wrapping(pt, config_scenes):
    # configuration 5's scene at its own fixed-point scale, 64 samples per pixel on a 256 x 256
    # image: no sum comes near 32 bits, and the statistic says so
    arrays, info = config_scenes(5, dragon_n_u=96, dragon_n_v=24)
    pt.upload_scene(arrays)
    W = H = 256
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=8, seed=info.seed, shade_mode=1, nee=1)
    pt.create_state(W, H, W * H * 4, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(64)
    s4, cnt = pt.read_accum()
    assert pt.stats().overflowed_sums == 0 and cnt.min() == 64
    # largest sum x (1024 / 64) stays below 2^32: configuration 5's 1024 spp fit too
    assert int(s4.max()) * 16 < 2 ** 32
    # a light 4e5 times brighter: sums reach the limit, stay there in any order of the adds (the
    # oracle, sequential, ends on the same bits) and the statistic is non-zero on both sides
    arrays1, info1 = config_scenes(1)
    hot = {k: v.copy() for k, v in arrays1.items()}
    hot["materials"]["emissive"] *= np.float32(4e5)
    pt.upload_scene(hot)
    W = H = 48
    src = host.make_sample_sources(W, H, 3)
    cam = host.default_camera(W, H)
    prm = pathtracer.default_params(max_bounces=4, seed=3)
    pt.create_state(W, H, 700, src, prm)
    pt.set_camera(cam)
    pt.stats_reset()
    pt.render_spp(24)
    s4, cnt = pt.read_accum()
    w4, wcnt, wst = po.Scene(hot).render(cam, src, W, H, po.default_params(max_bounces=4, seed=3), 0, 24)
    assert np.array_equal(s4, w4) and np.array_equal(cnt, wcnt)
    assert (s4 == 0xFFFFFFFF).sum() > 10 and pt.stats().overflowed_sums > 0 and wst.overflowed_sums > 0
    # two contexts' sums that overflow only when added: the group reduction saturates too
    with pathtracer.PathtracerGroup([0, 0]) as grp:
        grp.upload_scene(hot)
        grp.create_state(W, H, 700, src, prm)
        grp.set_camera(cam)
        grp.render_spp(24)
        grp.reduce()
        g4, gcnt = grp.read_accum()
    assert np.array_equal(g4, w4) and np.array_equal(gcnt, wcnt)
    # the sums as a caller's collective sees them: copied to device buffers clamped, loaded back, added to again —
    # what stood at the limit stays there, the statistic still reports it, and everything ends on the bits of one
    # uninterrupted render of as many samples (per-sample streams: sample numbers 0..23 then 24..35)
    hip = _loaded_hip_runtime()
    d4, dc = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d4), 16 * len(src)) == 0 and hip.hipMalloc(C.byref(dc), 4 * len(src)) == 0
    pt.copy_accum_to_device(d4.value, dc.value)
    assert hip.hipDeviceSynchronize() == 0
    b4, bc = np.zeros(4 * len(src), "<u4"), np.zeros(len(src), "<u4")
    assert hip.hipMemcpy(C.c_void_p(b4.ctypes.data), d4, b4.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    assert hip.hipMemcpy(C.c_void_p(bc.ctypes.data), dc, bc.nbytes, 2) == 0
    assert np.array_equal(b4, s4) and np.array_equal(bc, cnt)
    pt.reset_accum()
    pt.load_accum_from_device(d4.value, dc.value)
    assert pt.stats().overflowed_sums > 0
    pt.set_sample_range(24)
    pt.render_spp(12)
    t4, tcnt = pt.read_accum()
    x4, xcnt, _ = po.Scene(hot).render(cam, src, W, H, po.default_params(max_bounces=4, seed=3), 0, 36)
    assert np.array_equal(t4, x4) and np.array_equal(tcnt, xcnt)
    assert (t4 == 0xFFFFFFFF).sum() >= (s4 == 0xFFFFFFFF).sum()
    assert hip.hipFree(d4) == 0 and hip.hipFree(dc) == 0
