"""Base-colour and emissive textures (the reference carries the ids and binds nothing,
src/material.rs:15-18): the oracle's sampling rule against an independently written numpy
sampler (CPU), and the HIP path against the oracle, bit-exact, in both shading paths (GPU)."""
import numpy as np
import pytest

from raytracer_amd import ffi, host, pathtracer
from oracle import pyoracle as po


def checker(w, h, a=(255, 255, 255), b=(40, 80, 200), cell=2, seed=None):
    t = np.zeros((h, w), "<u4")
    for y in range(h):
        for x in range(w):
            c = a if ((x // cell) + (y // cell)) % 2 == 0 else b
            t[y, x] = c[0] | (c[1] << 8) | (c[2] << 16) | (255 << 24)
    if seed is not None:  # every texel different, so that a wrong texel cannot go unnoticed
        t = np.random.default_rng(seed).integers(0, 2 ** 24, size=(h, w), dtype=np.uint32).astype("<u4") | np.uint32(255 << 24)
    return t


def numpy_tex(tex, filt, tu, tv):
    """The sampling rule of mirt.h written from its text: repeat addressing, channel / 255,
    nearest, or bilinear with texel centres at (i + 0.5) / size.  float32 throughout."""
    f = np.float32
    h, w = tex.shape
    rgb = np.stack([(tex >> s) & 255 for s in (0, 8, 16)], -1).astype(f) / f(255)
    fu, fv = f(tu) - np.floor(f(tu)), f(tv) - np.floor(f(tv))
    if filt == ffi.FILTER_NEAREST:
        return rgb[min(int(fv * f(h)), h - 1), min(int(fu * f(w)), w - 1)]
    sx, sy = fu * f(w) - f(0.5), fv * f(h) - f(0.5)
    x0, y0 = np.floor(sx), np.floor(sy)
    tx, ty = f(sx - x0), f(sy - y0)
    x0, y0 = int(x0) % w, int(y0) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    top = rgb[y0, x0] * f(1 - tx) + rgb[y0, x1] * tx
    bot = rgb[y1, x0] * f(1 - tx) + rgb[y1, x1] * tx
    return top * f(1 - ty) + bot * ty


def panel_scene(filt, tex):
    """A rect that fills the view at z = 3, emissive through a texture, uv = (x, y) + 0.5."""
    s = host.Scene()
    rect = s.load_rect()
    s.set_planar_uv(rect, 0, 1, 1.0, 0.5)
    tid = s.add_texture(tex, filt)
    m = s.add_material(host.make_material(colour=(0, 0, 0, 1), emissive=(4, 2, 1, 1), emissive_texture=tid))
    s.spawn(host.make_transform(scale=(8, 8, 1), translation=(0, 0, 3)), m, rect)
    arrays = s.bind()
    s.close()
    return arrays


@pytest.mark.parametrize("filt", [ffi.FILTER_NEAREST, ffi.FILTER_BILINEAR])
def test_oracle_texture_rule_against_a_numpy_sampler(filt):
    tex = checker(7, 5, seed=3)
    arrays = panel_scene(filt, tex)
    assert arrays["textures"].tolist() == [(7, 5, 0, filt)] and len(arrays["texels"]) == 35
    assert arrays["vertices"]["uv"][:4, :2].tolist() == [[0, 0], [1, 0], [1, 1], [0, 1]]
    W = H = 48
    src = host.make_sample_sources(W, H, 5, shuffle=0)
    cam = host.default_camera(W, H)
    prm = po.default_params(max_bounces=1, seed=5, sky_radiance=0.0)
    sc = po.Scene(arrays)
    s4, cnt, st = sc.render(cam, src, W, H, prm, 0, 1)
    assert st.shade_events == W * H  # every camera ray hits the panel
    rays = sc.bounce_rays(cam, src, W, H, prm, 0, 0)
    d = rays["dir"].astype(np.float64)
    hit = d * (3.0 / d[:, 2:3])          # the panel is the plane z = 3
    tu, tv = hit[:, 0] / 8.0 + 0.5, hit[:, 1] / 8.0 + 0.5  # object-space x, y of the scaled rect, + 0.5
    emis = np.array([4, 2, 1], np.float32)
    want = np.stack([numpy_tex(tex, filt, tu[i], tv[i]) * emis * np.float32(1000) for i in range(len(src))])
    got = s4.reshape(-1, 4)[:, :3].astype(np.float64)
    # the oracle interpolates the vertex uvs with the hit's barycentrics, this test uses the hit point: a few
    # ulp apart, which moves a fixed-point sum by one unit — or, for a hit on a texel border with the nearest
    # filter, to the neighbouring texel.  Tolerance: 2 units of 1/1000, on all but 1 % of the pixels.
    close = np.all(np.abs(got - np.floor(want)) <= 2.0, axis=1)
    assert close.mean() > (0.99 if filt == ffi.FILTER_NEAREST else 0.999)


def textured_room():
    s = host.Scene()
    rect, cube = s.load_rect(), s.load_cube()
    suz = s.load_model_file(host.default_asset_dir() + "/suzanne.mirtmesh")
    s.set_planar_uv(rect, 0, 1, 1.0, 0.5)
    s.set_planar_uv(suz, 0, 2, 1.5, 0.25)
    s.set_planar_uv(cube, 0, 2, 2.0, 0.0)
    floor_tex = s.add_texture(checker(8, 8, cell=1), ffi.FILTER_BILINEAR)
    noise_tex = s.add_texture(checker(13, 9, seed=8), ffi.FILTER_NEAREST)
    glow_tex = s.add_texture(checker(4, 4, a=(255, 200, 120), b=(10, 10, 10), cell=1), ffi.FILTER_BILINEAR)
    floor = s.add_material(host.make_material(colour=(0.9, 0.9, 0.9, 1), colour_texture=floor_tex))
    skin = s.add_material(host.make_material(colour=(1.0, 0.8, 0.7, 1), metallic=0.3, roughness=0.4, colour_texture=noise_tex))
    mirror = s.add_material(host.make_material(colour=(0.9, 0.9, 0.9, 1), metallic=1.0, roughness=0.05, colour_texture=noise_tex))
    glass = s.add_material(host.make_material(colour=(1, 1, 1, 1), transmission=1.0, colour_texture=floor_tex))
    glow = s.add_material(host.make_material(colour=(0, 0, 0, 1), emissive=(6, 6, 6, 1), emissive_texture=glow_tex))
    plain_light = s.add_material(host.make_material(colour=(0, 0, 0, 1), emissive=(9, 9, 9, 1)))
    s.spawn(host.make_transform(scale=(5, 5, 1), rotation=(np.pi / 2, 0, 0), translation=(0, -1.0, 3.0)), floor, rect)
    s.spawn(host.make_transform(scale=(0.8, 0.8, 0.8), translation=(-0.5, -0.2, 3.0)), skin, suz)
    s.spawn(host.make_transform(scale=(0.5, 0.5, 0.5), translation=(0.9, -0.6, 2.6)), mirror, cube)
    s.spawn(host.make_transform(scale=(0.4, 0.4, 0.4), rotation=(0.3, 0.5, 0.0), translation=(0.2, -0.6, 2.0)), glass, cube)
    s.spawn(host.make_transform(scale=(1.5, 1.5, 1), translation=(0, 0.3, 4.5)), glow, rect)
    return s, plain_light, rect


@pytest.mark.gpu
def test_textured_scene_matches_the_oracle_in_both_shading_paths():
    s, plain_light, rect = textured_room()
    arrays = s.bind()
    assert len(arrays["textures"]) == 3 and arrays["materials"]["colour_texture"].max() == 2
    W, H = 96, 64
    src = host.make_sample_sources(W, H, 21)
    cam = host.default_camera(W, H)
    with pathtracer.Pathtracer(0) as pt:
        pt.upload_scene(arrays)
        sc = po.Scene(arrays)
        for mode, nee in ((0, 0), (1, 0)):
            kw = dict(max_bounces=5, seed=21, shade_mode=mode, nee=nee)
            pt.create_state(W, H, 2500, src, pathtracer.default_params(**kw))
            pt.set_camera(cam)
            pt.stats_reset()
            pt.render_spp(3)
            s4, cnt = pt.read_accum()
            w4, wcnt, wst = sc.render(cam, src, W, H, po.default_params(**kw), 0, 3)
            assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
            assert pt.stats().shade_events == wst.shade_events > W * H
            # the textures do change the picture
            flat = {k: v.copy() for k, v in arrays.items()}
            flat["materials"]["colour_texture"] = 0
            flat["materials"]["emissive_texture"] = 0
            f4, _, _ = po.Scene(flat).render(cam, src, W, H, po.default_params(**kw), 0, 3)
            assert not np.array_equal(f4, w4)
            # the per-frame call (three-kernel frame) reads them too
            pt.reset_accum()
            pt.step(9)
            s4, cnt = pt.read_accum()
            w4, wcnt, _ = sc.step_wavefront(cam, src, W, H, po.default_params(**kw), 0, 9, 2500)
            assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
        # light sampling: the textured emitter is on the light list (its texel is looked up at the sampled
        # point), next to an untextured one added through the instance-update call
        glow_inst = int(arrays["light_sources"][0])
        s.spawn(host.make_transform(scale=(1, 1, 1), rotation=(np.pi / 2, 0, 0), translation=(0, 1.8, 3.0)), plain_light, rect)
        lit = s.bind()
        s.close()
        assert len(lit["light_sources"]) == 2 and glow_inst in lit["light_sources"].tolist()
        assert lit["materials"]["emissive_texture"].max() == 3
        pt.update_instances(lit)  # textures and materials are instance-side data: no geometry upload
        kw = dict(max_bounces=5, seed=21, shade_mode=1, nee=1)
        w4, wcnt, wst = po.Scene(lit).render(cam, src, W, H, po.default_params(**kw), 0, 2)
        for stage in (0, 1):  # the materials in one kernel / in a kernel each behind a routing pass
            pt.create_state(W, H, 2500, src, pathtracer.default_params(stage_kernels=stage, **kw))
            pt.set_camera(cam)
            pt.stats_reset()
            pt.render_spp(2)
            s4, cnt = pt.read_accum()
            assert wst.connect_rays > 1000 and pt.stats().connect_rays == wst.connect_rays
            assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4), stage
        untextured = {k: v.copy() for k, v in lit.items()}
        untextured["materials"]["emissive_texture"] = 0
        u4, _, _ = po.Scene(untextured).render(cam, src, W, H, po.default_params(**kw), 0, 2)
        assert not np.array_equal(u4, w4)  # the light's texture is seen by the connect stage
        # a scene first uploaded without textures gets them through the update call (falls back to a full upload)
        bare = {k: v.copy() for k, v in lit.items() if k not in ("textures", "texels")}
        bare["materials"]["colour_texture"] = 0
        bare["materials"]["emissive_texture"] = 0
        pt.upload_scene(bare)
        pt.update_instances(lit)
        pt.create_state(W, H, 2500, src, pathtracer.default_params(**kw))
        pt.set_camera(cam)
        pt.render_spp(2)
        assert np.array_equal(pt.read_accum()[0], w4)
        # bad texture references are reported
        bad = {k: v.copy() for k, v in lit.items()}
        bad["materials"]["colour_texture"][0] = 9
        with pytest.raises(pathtracer.MirtError, match="texture"):
            pt.upload_scene(bad)
        bad = {k: v.copy() for k, v in lit.items()}
        bad["textures"]["first_texel"][2] = 10 ** 6
        with pytest.raises(pathtracer.MirtError, match="texels out of range"):
            pt.upload_scene(bad)
