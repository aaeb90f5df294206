"""CPU tests: pin the oracle and the host library against the known-answer
fixtures in tests/golden/ and against each other.

The reference ships no tests, golden vectors or fixtures for this path and
cannot be built here, so parity with it is UNPINNED; what is pinned is this
repository's restatement: values derived from the text of the reference's
formulas (RNG), literal geometry (rect, cube), independent re-implementations
(pure-Python builder vs C++ builder, brute force vs BVH, numpy float32
restatements of the shader functions written from the shader text vs the C
oracle) and invariants.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, hits_equal, random_rays
from raytracer_amd import ffi, host, pathtracer
from oracle import host_ref, pyoracle as po

KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))
GOLD = np.load(os.path.join(GOLDEN, "golden.npz"), allow_pickle=False)


# ------------------------------------------------------------------ RNG
def test_rng_matches_values_derived_from_reference_text():
    # shaders/random.slang:9-30; expected values computed with Python integers
    for key, rows in KAT["rng"].items():
        seed = [int(v) for v in key.split(",")]
        vals, st = po.rng_sequence(seed, len(rows))
        assert [int(b) for b in vals.view("<u4")] == [r["value_bits"] for r in rows]
        assert [int(v) for v in st] == rows[-1]["state"]
    first = KAT["rng"]["1000,1000,1000,1000"][0]
    assert first["state"] == [0x003E800F, 0x00003E80, 0x07C00003, 0x9FA59627]
    assert first["value_bits"] == 0x3F185B29


def test_rng_values_lie_in_unit_interval_inclusive():
    vals, _ = po.rng_sequence([1000, 2000, 3000, 4000], 4096)
    assert vals.min() >= 0.0 and vals.max() <= 1.0


def test_rng_stream_key_words_are_at_least_1000_and_distinct():
    seen = set()
    for source in range(64):
        for k in range(4):
            st = po.rng_init(0x5EED0001, source, k)
            assert (st >= 1000).all()
            seen.add(tuple(int(v) for v in st))
    assert len(seen) == 64 * 4


# ---------------------------------------------------------------- builders
def _nodes_of(scene, mesh_id):
    pos, nrm, faces, nodes = scene.mesh(mesh_id)
    recs = [[[float(v) for v in n["lb"][:3]], [float(v) for v in n["ub"][:3]], int(n["left"]), int(n["right"]),
             int(n["is_leaf"]), int(n["start"]), int(n["end"])] for n in nodes]
    return pos, nrm, faces, recs


def test_rect_is_a_single_leaf_root():
    s = host.Scene()
    pos, nrm, faces, recs = _nodes_of(s, s.load_rect())
    assert recs == KAT["bvh"]["rect"]["nodes"] == [[[-0.5, -0.5, 0.0], [0.5, 0.5, 0.0], 0, 0, 1, 0, 2]]
    assert faces.tolist() == [[0, 1, 2, 0], [0, 2, 3, 0]]


def test_cube_bvh_matches_python_builder_and_hand_traced_root_split():
    s = host.Scene()
    pos, nrm, faces, recs = _nodes_of(s, s.load_cube())
    assert recs == KAT["bvh"]["cube"]["nodes"]
    assert faces.tolist() == KAT["bvh"]["cube"]["faces"]
    # all extents are equal, so axis 0 is chosen (src/bvh.rs:82-88 uses strict >);
    # centroids add positions[0] = (0.5,-0.5,-0.5) as a fourth term (src/blas.rs:33-37)
    # -> x in {2/3, -1/3, 0, 1/3}, split = 1/6.  Six triangles lie below the split,
    # but the one under the cursors' meeting point is never tested (src/bvh.rs:105-112)
    # and lands right: 5 left, 7 right.
    root = recs[0]
    assert root[2:] == [1, 0, 0, 0, 12]
    assert recs[1][5:] == [0, 5] and recs[2][5:] == [5, 12]
    # skip links: left -> right sibling, right -> parent's next, root -> 0 (src/bvh.rs:159-172)
    assert recs[1][3] == 2 and recs[2][3] == 0


def test_suzanne_builder_matches_python_restatement():
    s = host.Scene()
    mid = s.load_model_file(os.path.join(GOLDEN, "suzanne.mirtmesh"))
    pos, nrm, faces, recs = _nodes_of(s, mid)
    assert len(pos) == KAT["assets"]["suzanne"]["positions"] and len(faces) == KAT["assets"]["suzanne"]["triangles"]
    # rebuild from the raw model with the pure-Python restatement
    raw = np.fromfile(os.path.join(GOLDEN, "suzanne.mirtmesh"), dtype="<u4")
    n_pos, n_nrm, n_idx = (int(v) for v in raw[2:5])
    fl = raw[5:].view("<f4")
    positions = fl[:n_pos].reshape(-1, 3)
    indices = raw[5 + n_pos + n_nrm: 5 + n_pos + n_nrm + n_idx].reshape(-1, 3)
    ppos, pfaces = host_ref.from_model(positions, indices)
    assert np.array_equal(np.array(ppos, "<f4").view("<u4"), pos.view("<u4"))
    pnodes, pfaces = host_ref.build_blas(ppos, pfaces)
    assert [list(map(int, f)) for f in pfaces] == faces.tolist()
    got = host_ref.nodes_as_records(pnodes)
    assert len(got) == len(recs)
    for (lb, ub, left, skip, leaf, start, end), r in zip(got, recs):
        assert [list(lb), list(ub), left, skip, leaf, start, end] == r


def test_standalone_host_builder_and_builder_callback():
    # mirth_build_blas_host on raw positions against the pure-Python builder, and the
    # builder callback of the host model (the hook the device builder plugs into) fed
    # with that same host builder: identical scene buffers
    import ctypes as C
    rng = np.random.default_rng(3)
    pos = np.ones((40, 4), "<f4")
    pos[:, :3] = rng.uniform(-1.0, 1.0, (40, 3))
    faces = np.zeros((150, 4), "<u4")
    faces[:, :3] = rng.integers(0, 40, (150, 3))
    faces[60:75] = faces[60]  # coincident triangles: a leaf the partition cannot split
    got_faces, got_nodes = host.build_blas(pos, faces)
    pnodes, pfaces = host_ref.build_blas([p for p in pos], faces.tolist())
    assert got_faces.tolist() == [list(map(int, f)) for f in pfaces]
    want = host_ref.nodes_as_records(pnodes)
    assert len(want) == len(got_nodes)
    for (lb, ub, left, skip, leaf, start, end), n in zip(want, got_nodes):
        assert (list(lb), list(ub)) == ([float(x) for x in n["lb"][:3]], [float(x) for x in n["ub"][:3]])
        assert (left, skip, leaf, start, end) == (n["left"], n["right"], n["is_leaf"], n["start"], n["end"])

    L = ffi.host_lib()
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                        C.POINTER(C.c_uint32))
    calls = []

    def via_host(user, p, n_p, f, n_f, nodes, cap, n_nodes):
        calls.append(n_f)
        return L.mirth_build_blas_host(p, n_p, f, n_f, nodes, cap, n_nodes)

    thunk = proto(via_host)
    want_arrays, _ = host.build_config_scene(1)
    s = host.Scene()
    s._check(L.mirth_set_blas_builder(s._h, C.cast(thunk, C.c_void_p), None), "set_blas_builder")
    s.build_config(1)
    got_arrays = s.bind()
    s.close()
    assert sorted(calls) == [2, 968]  # rect and suzanne went through the callback
    for key in ("indices", "blas_nodes", "tlas_nodes", "vertices"):
        assert np.asarray(got_arrays[key]).tobytes() == np.asarray(want_arrays[key]).tobytes(), key
    # a failing callback fails the load instead of leaving a half-built mesh behind
    bad = proto(lambda *a: 7)
    s = host.Scene()
    s._check(L.mirth_set_blas_builder(s._h, C.cast(bad, C.c_void_p), None), "set_blas_builder")
    with pytest.raises(RuntimeError):
        s.load_cube()
    s.close()


# ------------------------------------------- second restatement, in numpy
def _f32(x):
    return np.float32(x)


def _fma(a, b, c):
    """round_f32(a * b + c) with the product and the sum exact (rationals)."""
    from fractions import Fraction
    return np.float32(float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))))


def _dot3(a, b):  # the arithmetic contract: fma(az, bz, fma(ay, by, ax * bx))
    return _fma(a[2], b[2], _fma(a[1], b[1], _f32(a[0]) * _f32(b[0])))


def _cross3(a, b):  # cross.x = fma(ay, bz, -(az * by)), ...
    return [_fma(a[1], b[2], -(_f32(a[2]) * _f32(b[1]))), _fma(a[2], b[0], -(_f32(a[0]) * _f32(b[2]))),
            _fma(a[0], b[1], -(_f32(a[1]) * _f32(b[0])))]


def test_camera_rays_against_a_numpy_restatement_of_the_shader_text():
    # spawnSample (shaders/sample.slang:86-98) written a second time, in numpy float32 scalars from
    # the shader text and the arithmetic contract, against the C oracle's camera rays
    arrays, info = host.build_config_scene(1)
    W, H = 48, 32
    src = host.make_sample_sources(W, H, 77)
    cam = host.default_camera(W, H)
    cam["position"][0] = (0.25, -0.5, 0.125)
    cam["focal_length"] = 1.5
    prm = po.default_params(max_bounces=4, seed=0xABCDEF)
    sc = po.Scene(arrays)
    for sample_no in (0, 5):
        rays = sc.bounce_rays(cam, src, W, H, prm, sample_no, 0)
        fwd = [_f32(v) for v in cam["forward"][0]]
        up = [_f32(v) for v in cam["up"][0]]
        dx, dy = _f32(cam["dims"][0][0]), _f32(cam["dims"][0][1])
        focal = _f32(cam["focal_length"][0])
        right = _cross3(fwd, up)
        top_left = [(fwd[k] * focal + up[k] * dy) - right[k] * dx for k in range(3)]
        for i in range(0, len(src), 97):
            vals, _ = po.rng_sequence(po.rng_init(prm.seed, i, sample_no), 2)
            jx, jy = _f32(vals[0]) / _f32(W), _f32(vals[1]) / _f32(H)  # x takes the first draw
            sx, sy = _f32(src["screen_pos"][i][0]) + jx, _f32(src["screen_pos"][i][1]) + jy
            d = [top_left[k] + ((((up[k] * _f32(-2.0)) * dy) * sy) + (((right[k] * _f32(2.0)) * dx) * sx)) for k in range(3)]
            inv_len = _f32(1.0) / np.sqrt(_dot3(d, d))
            want = np.array([d[k] * inv_len for k in range(3)], "<f4")
            assert np.array_equal(rays["dir"][i].view("<u4"), want.view("<u4")), (sample_no, i)
            assert np.array_equal(rays["pos"][i], cam["position"][0])
            assert rays["last_inst"][i] == rays["last_prim"][i] == 0x7FFFFFFF  # int.maxValue


def test_tone_map_and_pixel_packing_against_a_numpy_restatement_of_the_shader_text():
    # acesToneMap + packRgb (shaders/sample.slang:18-33,52-56; shaders/common.slang:177-183) in numpy
    # float32, on the mean over the FINAL count (the documented difference from the reference)
    rng = np.random.default_rng(4)
    W, H = 16, 8
    src = host.make_sample_sources(W, H, 3)
    n = W * H
    cnt = rng.integers(0, 9, n).astype("<u4")
    s4 = np.zeros((n, 4), "<u4")
    s4[:, :3] = (rng.random((n, 3)) ** 3 * 40000.0 * np.maximum(cnt, 1)[:, None]).astype("<u4")
    prm = po.default_params()
    got = po.resolve(s4.reshape(-1).copy(), cnt, src, W, H, prm)
    m1 = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83770]], "<f4")
    m2 = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]], "<f4")
    sat = lambda x: _f32(0.0) if x < 0 else (_f32(1.0) if x > 1 else x)
    with np.errstate(all="ignore"):
        for i in range(n):
            ox, oy = int(src["out_pos"][i][0]), int(src["out_pos"][i][1])
            if cnt[i] == 0:
                assert got[ox + oy * W] == 0
                continue
            denom = _f32(1000.0) * _f32(cnt[i])
            rad = [(_f32(s4[i][k]) / denom) * _f32(prm.exposure) for k in range(3)]
            v = [m1[k][0] * rad[0] + m1[k][1] * rad[1] + m1[k][2] * rad[2] for k in range(3)]
            c = [(v[k] * (v[k] + _f32(0.0245786)) - _f32(0.000090537)) /
                 (v[k] * (_f32(0.983729) * v[k] + _f32(0.4329510)) + _f32(0.238081)) for k in range(3)]
            o = [sat(m2[k][0] * c[0] + m2[k][1] * c[1] + m2[k][2] * c[2]) for k in range(3)]
            r, g, b = (int(np.uint32(o[k] * _f32(255.0))) for k in range(3))
            assert got[ox + oy * W] == (b << 16) | (g << 8) | r, i  # .bgr swizzle: red in the low byte


def test_triangle_test_against_a_numpy_restatement_of_the_shader_text():
    # rayTriIntersect (shaders/ray_extend.slang:53-92) a second time in numpy float32, on the rect
    # mesh under an identity transform (object space = world space): hit triangle, t bits and
    # front_face of the C oracle's traversal for rays that clearly hit or clearly miss
    s = host.Scene()
    rect = s.load_rect()
    mat = s.add_material(host.default_material())
    s.spawn(host.make_transform(), mat, rect)
    arrays = s.bind()
    pos4, _, faces, _ = s.mesh(rect)
    s.close()
    sc = po.Scene(arrays)
    rng = np.random.default_rng(21)
    n = 400
    rays = np.zeros(n, ffi.Ray)
    target = rng.uniform(-0.75, 0.75, (n, 2))
    target = target[(np.abs(np.abs(target) - 0.5) > 0.05).all(axis=1)]  # stay off the rect's edges
    n = len(target)
    rays = rays[:n]
    origin = np.c_[rng.uniform(-1.0, 1.0, (n, 2)), rng.choice([-2.0, 1.5], n)]
    d = np.c_[target, np.zeros(n)] - origin
    rays["pos"] = origin.astype("<f4")
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rays["last_inst"] = rays["last_prim"] = ffi.NO_PRIM
    got = sc.trace_closest(rays)
    hits = 0
    for i in range(n):
        pos = [_f32(v) for v in rays["pos"][i]]
        dr = [_f32(v) for v in rays["dir"][i]]
        t, prim = np.float32(3.4028234663852886e38), None
        for p, face in enumerate(faces):
            p0, p1, p2 = ([_f32(v) for v in pos4[face[k]][:3]] for k in range(3))
            e1 = [p1[k] - p0[k] for k in range(3)]
            e2 = [p2[k] - p0[k] for k in range(3)]
            q = _cross3(dr, e2)
            alpha = _dot3(e1, q)
            if -_f32(10e-8) < alpha < _f32(10e-8):
                continue
            f = _f32(1.0) / alpha
            sv = [pos[k] - p0[k] for k in range(3)]
            u = f * _dot3(sv, q)
            if u < 0:
                continue
            r = _cross3(sv, e1)
            v = f * _dot3(dr, r)
            if v < 0 or u + v > 1:
                continue
            t2 = f * _dot3(e2, r)
            if t2 > t or t2 < 0:
                continue
            t, prim = t2, p  # an equal t replaces the earlier hit
        inside = bool((np.abs(target[i]) < 0.5).all())
        assert (prim is not None) == inside, i
        if prim is None:
            assert got["instance_id"][i] == 0xFFFFFFFF
        else:
            hits += 1
            assert got["instance_id"][i] == 0 and got["triangle_id"][i] == prim, i
            assert got["t"][i].view("<u4") == np.float32(t).view("<u4"), i
            # the rect's normal is +z; front_face = dot(normal, dir) < 0
            assert got["front_face"][i] == (1 if dr[2] < 0 else 0), i
    assert 50 < hits < n - 50


def _normalize3(a):
    inv = _f32(1.0) / np.sqrt(_dot3(a, a))
    return [a[k] * inv for k in range(3)]


def _camera_dir(cam, src_row, W, H, r1, r2):
    fwd = [_f32(v) for v in cam["forward"][0]]
    up = [_f32(v) for v in cam["up"][0]]
    dx, dy = _f32(cam["dims"][0][0]), _f32(cam["dims"][0][1])
    right = _cross3(fwd, up)
    top_left = [(fwd[k] * _f32(cam["focal_length"][0]) + up[k] * dy) - right[k] * dx for k in range(3)]
    sx = _f32(src_row["screen_pos"][0]) + _f32(r1) / _f32(W)
    sy = _f32(src_row["screen_pos"][1]) + _f32(r2) / _f32(H)
    d = [top_left[k] + ((((up[k] * _f32(-2.0)) * dy) * sy) + (((right[k] * _f32(2.0)) * dx) * sx)) for k in range(3)]
    return _normalize3(d)


@pytest.mark.parametrize("flip", [False, True])
def test_uber_shading_against_a_numpy_restatement_of_the_shader_text(flip):
    # One bounce of shadeMain (shaders/shade.slang:19-224) — emission, the dead draws, the cosine
    # sample, the glTF metal-rough BRDF, the throughput update — a second time in numpy float32:
    # a camera ray hits a big rect (seen from its back or, flipped in z, from its front), is shaded once
    # and leaves for the sky, so the sample's radiance is throughput x sky and its fixed-point sum
    # must equal the C oracle's.
    PI = _f32(3.14159265)
    W, H = 12, 8
    rng = np.random.default_rng(8 + flip)
    mats = []
    for _ in range(5):
        mats.append(dict(colour=tuple(rng.uniform(0.1, 1.0, 3)) + (1.0,), metallic=float(rng.choice([0.0, 1.0, rng.random()])),
                         roughness=float(rng.uniform(0.05, 1.0)), ior=float(rng.uniform(1.1, 2.0)),
                         transmission=float(rng.choice([0.0, 1.0, rng.random()]))))
    src = host.make_sample_sources(W, H, 5)
    cam = host.default_camera(W, H)
    sz = -1.0 if flip else 1.0
    for mi, mk in enumerate(mats):
        s = host.Scene()
        rect = s.load_rect()
        m = host.make_material(**mk)
        mid = s.add_material(m)
        s.spawn(host.make_transform(scale=(16.0, 16.0, sz), translation=(0.0, 0.0, 3.0)), mid, rect)
        arrays = s.bind()
        pos4, nrm4, faces, _ = s.mesh(rect)
        s.close()
        prm = po.default_params(max_bounces=2, seed=100 + mi)
        got4, gotc, _ = po.Scene(arrays).render(cam, src, W, H, prm, 0, 1)
        colour = [_f32(v) for v in mk["colour"][:3]]
        metallic, rough, ior, trans = (_f32(mk[k]) for k in ("metallic", "roughness", "ior", "transmission"))
        with np.errstate(all="ignore"):
            for i in range(0, len(src), 7):
                draws, _ = po.rng_sequence(po.rng_init(prm.seed, i, 0), 200)
                draws = [_f32(v) for v in draws]
                wdir = _camera_dir(cam, src[i], W, H, draws[0], draws[1])
                at = [2]

                def nxt():
                    at[0] += 1
                    return draws[at[0] - 1]
                # object space: M^-1 = S^-1 T^-1 with scale (16, 16, sz), translation (0, 0, 3): exact in f32
                opos = [_f32(0.0), _f32(0.0), _f32(-3.0) * _f32(1.0 / sz)]
                odir = [wdir[0] * _f32(0.0625), wdir[1] * _f32(0.0625), wdir[2] * _f32(1.0 / sz)]
                hit = None
                t = np.float32(3.4028234663852886e38)
                for p, face in enumerate(faces):
                    p0, p1, p2 = ([_f32(v) for v in pos4[face[k]][:3]] for k in range(3))
                    e1 = [p1[k] - p0[k] for k in range(3)]
                    e2 = [p2[k] - p0[k] for k in range(3)]
                    q = _cross3(odir, e2)
                    alpha = _dot3(e1, q)
                    if -_f32(10e-8) < alpha < _f32(10e-8):
                        continue
                    f = _f32(1.0) / alpha
                    sv = [opos[k] - p0[k] for k in range(3)]
                    u = f * _dot3(sv, q)
                    if u < 0:
                        continue
                    r = _cross3(sv, e1)
                    v = f * _dot3(odir, r)
                    if v < 0 or u + v > 1:
                        continue
                    t2 = f * _dot3(e2, r)
                    if t2 > t or t2 < 0:
                        continue
                    t, hit = t2, (p, u, v)
                assert hit is not None  # the rect fills the view
                p, u, v = hit
                w = _f32(1.0) - u - v
                n0, n1, n2 = ([_f32(x) for x in nrm4[faces[p][k]][:3]] for k in range(3))
                nobj = [_fma(n2[k], v, _fma(n1[k], u, n0[k] * w)) for k in range(3)]
                normal = _normalize3([nobj[0] * _f32(16.0), nobj[1] * _f32(16.0), nobj[2] * _f32(sz)])  # M, not M^-T
                ff = _dot3(normal, wdir) < 0
                assert ff == flip  # unflipped, the rect's +z normal points away from the camera
                n = [x * (_f32(1.0) if ff else _f32(-1.0)) for x in normal]
                wo = wdir  # the incoming direction, not negated

                def disk():
                    while True:
                        px = _f32(2.0) * nxt() - _f32(1.0)
                        py = _f32(2.0) * nxt() - _f32(1.0)
                        if not np.sqrt(_fma(py, py, px * px)) > 1:
                            return px, py

                def cosine_sample():
                    dx, dy = disk()
                    z = np.sqrt(max(_f32(0.0), _f32(1.0) - dx * dx - dy * dy))
                    temp = [_f32(0), _f32(1), _f32(0)] if abs(n[0]) > _f32(0.9) else [_f32(1), _f32(0), _f32(0)]
                    t1 = _normalize3(_cross3(n, temp))
                    t2_ = _cross3(n, t1)
                    nn = _normalize3(n)
                    return [_fma(nn[k], z, _fma(t2_[k], dy, t1[k] * dx)) for k in range(3)]
                cosine_sample()  # diffuse_sample: unused, its draws are not
                while True:      # metallicSample -> unitSphereSample: three draws per try
                    r3 = [nxt(), nxt(), nxt()]
                    if not np.sqrt(_dot3(r3, r3)) > 1:
                        break
                wi = cosine_sample()
                pdf = abs(_dot3(wi, n)) / PI
                heav = lambda x: _f32(1.0) if x > 0 else _f32(0.0)
                pow5 = lambda x: np.float32(np.nan) if x < 0 else ((x * x) * (x * x)) * x
                alpha_r = rough * rough
                a2 = alpha_r * alpha_r
                h = _normalize3([wo[k] + wi[k] for k in range(3)])
                fres = pow5(_f32(1.0) - abs(_dot3(wo, h)))
                nh = _dot3(n, h)
                d = a2 * heav(nh)
                d = d / (PI * ((nh * nh) * (a2 - _f32(1.0)) + _f32(1.0)) ** 2)
                nwi, nwo = _dot3(n, wi), _dot3(n, wo)
                vis = heav(_dot3(h, wi)) / (abs(nwi) + np.sqrt(a2 + (_f32(1.0) - a2) * (nwi * nwi)))
                vis = vis * (heav(_dot3(h, wo)) / (abs(nwo) + np.sqrt(a2 + (_f32(1.0) - a2) * (nwo * nwo))))
                spec = vis * d
                k2 = _f32(2.0) * nwi
                ht = _normalize3([(wo[k] - n[k] * k2) + wi[k] for k in range(3)])
                nht = _dot3(n, ht)
                dt = a2 * heav(nht)
                dt = dt / (PI * ((nht * nht) * (a2 - _f32(1.0)) + _f32(1.0)) ** 2)
                vt = heav(_dot3(ht, wi) / nwi)
                vt = vt * heav(_dot3(ht, wo) / nwo)
                vt = vt / (abs(nwi) + np.sqrt(a2 + (_f32(1.0) - a2)) * (nwi * nwi))
                vt = vt / (abs(nwo) + np.sqrt(a2 + (_f32(1.0) - a2)) * (nwo * nwo))
                btdf = vt * dt
                mix = lambda a, b, mm: (_f32(1.0) - mm) * a + mm * b
                f0 = ((_f32(1.0) - ior) / (_f32(1.0) + ior)) ** 2
                fr = f0 + (_f32(1.0) - f0) * fres
                ndotwi = abs(_dot3(n, wi))
                for c in range(3):
                    base = mix(colour[c] * (_f32(1.0) / PI), colour[c] * btdf, trans)
                    dielectric = mix(base, spec, fr)
                    metal = spec * (colour[c] + (_f32(1.0) - colour[c]) * fres)
                    brdf = mix(dielectric, metal, metallic)
                    thr = _f32(1.0) * (brdf * ndotwi * _f32(1.0) / pdf)
                    rad = _f32(0.0) + thr * _f32(10.0)  # the sky ends the path
                    if np.isnan(rad) or np.isinf(rad):
                        rad = _f32(0.0)
                    x = rad * _f32(1000.0)
                    want = 0 if not x >= 0 else (4294967040 if x >= 4294967040.0 else int(np.uint32(x)))
                    assert int(got4[4 * i + c]) == want, (mi, i, c)
                assert gotc[i] == 1


def test_traversal_and_visit_counts_against_a_python_restatement_of_the_shader_text():
    # rayBoxIntersect, blasFirstHit and tlasFirstHit (shaders/ray_extend.slang:94-215) a second time,
    # as Python loops over the flat scene buffers with numpy float32 scalars: hit ids, t bits and the
    # node / instance / triangle visit counts the bandwidth model is built on.  Translation-only
    # instances keep the instance matrices exact without trigonometry.
    s = host.Scene()
    suz = s.load_model_file(os.path.join(GOLDEN, "suzanne.mirtmesh"))
    cube, rect = s.load_cube(), s.load_rect()
    mat = s.add_material(host.default_material())
    places = [(suz, (0.0, 0.0, 3.0)), (cube, (1.25, 0.5, 4.0)), (rect, (-1.0, 0.25, 3.5)), (cube, (-0.5, -1.0, 2.5))]
    for mesh, tr in places:
        s.spawn(host.make_transform(translation=tr), mat, mesh)
    arrays = s.bind()
    s.close()
    sc = po.Scene(arrays)
    rays = random_rays(60, 12, centre=(0.0, 0.0, 3.0), spread=1.5)
    rays["last_inst"][:10] = 0  # some rays carry an excluded triangle of instance 0
    rays["last_prim"][:10] = np.arange(10) * 7
    got, gc = sc.trace_closest(rays, want_counts=True)

    inst, offs = arrays["instances"], arrays["geometry_offsets"]
    verts, idx = arrays["vertices"], arrays["indices"]
    bn, tn, t2i, xf = arrays["blas_nodes"], arrays["tlas_nodes"], arrays["tlas_to_instances"], arrays["transforms"]
    FMAX = np.float32(3.4028234663852886e38)
    counts = dict(tlas=0, inst=0, blas=0, tri=0)
    fmax = lambda a, b: b if np.isnan(a) else (a if np.isnan(b) else (a if a > b else b))
    fmin = lambda a, b: b if np.isnan(a) else (a if np.isnan(b) else (a if a < b else b))

    def box(pos, dinv, node, t):
        tmin, tmax = -FMAX, t
        for d in range(3):
            sign = dinv[d] >= 0
            lo, hi = _f32(node["lb"][d]), _f32(node["ub"][d])
            bmin, bmax = (lo, hi) if sign else (hi, lo)
            tmin = fmax((bmin - pos[d]) * dinv[d], tmin)
            tmax = fmin((bmax - pos[d]) * dinv[d], tmax)
        return tmin <= tmax, tmin, tmax

    def tri(pos, dr, face, voff, t):
        p0, p1, p2 = ([_f32(x) for x in verts[voff + int(face["v"][k])]["position"][:3]] for k in range(3))
        e1 = [p1[k] - p0[k] for k in range(3)]
        e2 = [p2[k] - p0[k] for k in range(3)]
        q = _cross3(dr, e2)
        alpha = _dot3(e1, q)
        if -_f32(10e-8) < alpha < _f32(10e-8):
            return None
        f = _f32(1.0) / alpha
        sv = [pos[k] - p0[k] for k in range(3)]
        u = f * _dot3(sv, q)
        if u < 0:
            return None
        r = _cross3(sv, e1)
        v = f * _dot3(dr, r)
        if v < 0 or u + v > 1:
            return None
        t2 = f * _dot3(e2, r)
        if t2 > t or t2 < 0:
            return None
        return t2

    def blas(pos, dr, iid, last_inst, last_prim, t):
        off = offs[int(inst[iid]["geometry"])]
        dinv = [_f32(1.0) / dr[k] for k in range(3)]
        cur, best = 0, None
        while True:
            node = bn[cur + int(off["blas_node"])]
            counts["blas"] += 1
            hit, _, _ = box(pos, dinv, node, t)
            cur = int(node["right"]) if (node["is_leaf"] == 1 or not hit) else int(node["left"])
            if hit and node["is_leaf"] == 1:
                for p in range(int(node["start"]), int(node["end"])):
                    if p == last_prim and iid == last_inst:
                        continue
                    counts["tri"] += 1
                    t2 = tri(pos, dr, idx[p + int(off["index"])], int(off["vertex"]), t)
                    if t2 is not None:
                        t, best = t2, p
            if cur == 0:
                return best, t

    with np.errstate(all="ignore"):
        for i in range(len(rays)):
            pos = [_f32(x) for x in rays["pos"][i]]
            dr = [_f32(x) for x in rays["dir"][i]]
            li, lp = int(rays["last_inst"][i]), int(rays["last_prim"][i])
            dinv = [_f32(1.0) / dr[k] for k in range(3)]
            t, cur, best = FMAX, 0, None
            while True:
                node = tn[cur]
                counts["tlas"] += 1
                hit, tmin, tmax = box(pos, dinv, node, t)
                hit = hit and (tmax >= 0 or tmin >= 0)
                cur = int(node["right"]) if (node["is_leaf"] == 1 or not hit) else int(node["left"])
                if hit and node["is_leaf"] == 1:
                    for slot in range(int(node["start"]), int(node["end"])):
                        iid = int(t2i[slot])
                        counts["inst"] += 1
                        tr = [_f32(x) for x in xf[int(inst[iid]["transform"])]["translation"][:3]]
                        opos = [pos[k] + (-tr[k]) for k in range(3)]  # M^-1 of a pure translation
                        p, t2 = blas(opos, dr, iid, li, lp, t)
                        if p is not None:
                            t, best = t2, (iid, p)
                if cur == 0:
                    break
            if best is None:
                assert got["instance_id"][i] == 0xFFFFFFFF, i
            else:
                assert (int(got["instance_id"][i]), int(got["triangle_id"][i])) == best, i
                assert got["t"][i].view("<u4") == np.float32(t).view("<u4"), i
    assert (gc.tlas_nodes, gc.instance_visits, gc.blas_nodes, gc.triangles) == \
           (counts["tlas"], counts["inst"], counts["blas"], counts["tri"])
    assert (got["instance_id"] != 0xFFFFFFFF).sum() > 10


def _rect_hit(wdir, sz, pos4, nrm4, faces):
    """Camera ray from the origin against the rect scaled (16, 16, sz) at (0, 0, 3): rayTriIntersect in
    object space and the hit record of extensionMain (position, normal, front_face), in numpy float32."""
    opos = [_f32(0.0), _f32(0.0), _f32(-3.0) * _f32(1.0 / sz)]
    odir = [wdir[0] * _f32(0.0625), wdir[1] * _f32(0.0625), wdir[2] * _f32(1.0 / sz)]
    hit, t = None, np.float32(3.4028234663852886e38)
    for p, face in enumerate(faces):
        p0, p1, p2 = ([_f32(v) for v in pos4[face[k]][:3]] for k in range(3))
        e1 = [p1[k] - p0[k] for k in range(3)]
        e2 = [p2[k] - p0[k] for k in range(3)]
        q = _cross3(odir, e2)
        alpha = _dot3(e1, q)
        if -_f32(10e-8) < alpha < _f32(10e-8):
            continue
        f = _f32(1.0) / alpha
        sv = [opos[k] - p0[k] for k in range(3)]
        u = f * _dot3(sv, q)
        if u < 0:
            continue
        r = _cross3(sv, e1)
        v = f * _dot3(odir, r)
        if v < 0 or u + v > 1:
            continue
        t2 = f * _dot3(e2, r)
        if t2 > t or t2 < 0:
            continue
        t, hit = t2, (p, u, v, p0, e1, e2)
    assert hit is not None
    p, u, v, p0, e1, e2 = hit
    w = _f32(1.0) - u - v
    n0, n1, n2 = ([_f32(x) for x in nrm4[faces[p][k]][:3]] for k in range(3))
    nobj = [_fma(n2[k], v, _fma(n1[k], u, n0[k] * w)) for k in range(3)]
    pobj = [_fma(e2[k], v, _fma(e1[k], u, p0[k])) for k in range(3)]
    # M = T S: rows (16, 0, 0 | 0), (0, 16, 0 | 0), (0, 0, sz | 3), as fma(m2, z, fma(m1, y, m0 * x)) + m3
    hpos = [_fma(_f32(0), pobj[2], _fma(_f32(0), pobj[1], _f32(16.0) * pobj[0])) + _f32(0.0),
            _fma(_f32(0), pobj[2], _fma(_f32(16.0), pobj[1], _f32(0) * pobj[0])) + _f32(0.0),
            _fma(_f32(sz), pobj[2], _fma(_f32(0), pobj[1], _f32(0) * pobj[0])) + _f32(3.0)]
    normal = _normalize3([nobj[0] * _f32(16.0), nobj[1] * _f32(16.0), nobj[2] * _f32(sz)])
    ff = bool(_dot3(normal, wdir) < 0)
    return hpos, normal, ff


@pytest.mark.parametrize("kind", ["lambertian", "metallic", "dielectric"])
def test_per_material_kernels_against_a_numpy_restatement(kind):
    # One bounce of the per-material path without light sampling — the scatter of
    # archive/lambertian.slang:26-35 inside process_material (archive/material.slang:17-139),
    # archive/metallic.slang:65-96, archive/dielectric.slang:34-95 with the documented repairs — a
    # second time in numpy float32: the bounce ray (origin and direction bits) and, the ray then
    # leaving for the sky, the sample's fixed-point sums.
    W, H = 12, 8
    src = host.make_sample_sources(W, H, 6)
    cam = host.default_camera(W, H)
    rng = np.random.default_rng({"lambertian": 1, "metallic": 2, "dielectric": 3}[kind])
    for flip in (False, True):
        sz = -1.0 if flip else 1.0
        colour = tuple(float(x) for x in rng.uniform(0.2, 0.9, 3)) + (1.0,)
        mk = dict(colour=colour, metallic=1.0 if kind == "metallic" else 0.0, roughness=0.125,
                  ior=float(rng.uniform(1.2, 1.8)), transmission=1.0 if kind == "dielectric" else 0.0)
        s = host.Scene()
        rect = s.load_rect()
        mid = s.add_material(host.make_material(**mk))
        s.spawn(host.make_transform(scale=(16.0, 16.0, sz), translation=(0.0, 0.0, 3.0)), mid, rect)
        arrays = s.bind()
        pos4, nrm4, faces, _ = s.mesh(rect)
        s.close()
        prm = po.default_params(max_bounces=3, seed=300 + flip, shade_mode=1)
        sc = po.Scene(arrays)
        got4, gotc, _ = sc.render(cam, src, W, H, prm, 0, 1)
        rays = sc.bounce_rays(cam, src, W, H, prm, 0, 1)
        albedo = [_f32(v) for v in colour[:3]]
        with np.errstate(all="ignore"):
            for i in range(0, len(src), 5):
                draws, _ = po.rng_sequence(po.rng_init(prm.seed, i, 0), 8)
                draws = [_f32(v) for v in draws]
                wdir = _camera_dir(cam, src[i], W, H, draws[0], draws[1])
                hpos, normal, ff = _rect_hit(wdir, sz, pos4, nrm4, faces)
                nf = [x * (_f32(1.0) if ff else _f32(-1.0)) for x in normal]
                a, b, c = draws[2], draws[3], draws[4]
                unit = lambda: _normalize3([_f32(1.0) - _f32(2.0) * a, _f32(1.0) - _f32(2.0) * b, _f32(1.0) - _f32(2.0) * c])
                thr = [_f32(1.0)] * 3
                if kind == "lambertian":
                    wo = [-x for x in wdir]
                    scv = [nf[k] + unit()[k] for k in range(3)]
                    wi = _normalize3(nf if np.sqrt(_dot3(scv, scv)) < _f32(1e-6) else scv)
                    pdf = max(_f32(0.0), _dot3(wi, nf)) / _f32(3.1415)
                    assert pdf > 0
                    lit = not (_dot3(nf, wi) <= 0 or _dot3(nf, wo) <= 0)
                    for k in range(3):
                        x = thr[k] * (albedo[k] / _f32(3.1415) if lit else _f32(0.0))
                        x = x * _dot3(wi, nf)
                        x = x * ((_f32(1.0) * pdf) / (_f32(1.0) * pdf + _f32(1.0) * _f32(0.0)))
                        thr[k] = x / (_f32(1.0) * pdf)
                    new_pos = [hpos[k] + nf[k] * _f32(1e-3) for k in range(3)]
                    new_dir = wi
                elif kind == "metallic":
                    thr = [thr[k] * albedo[k] for k in range(3)]
                    k2 = _f32(2.0) * _dot3(wdir, nf)
                    refl = [wdir[k] - nf[k] * k2 for k in range(3)]
                    fz = unit()
                    scv = [refl[k] + fz[k] * _f32(0.125) for k in range(3)]
                    new_dir = _normalize3(refl if np.sqrt(_dot3(scv, scv)) < _f32(1e-2) else scv)
                    new_pos = [hpos[k] + nf[k] * _f32(0.005) for k in range(3)]
                else:
                    ir = _f32(mk["ior"])
                    ratio = _f32(1.0) / ir if ff else ir
                    ud = _normalize3(wdir)
                    cos_t = min(-_dot3(ud, nf), _f32(1.0))
                    sin_t = np.sqrt(max(_f32(0.0), _f32(1.0) - cos_t * cos_t))
                    r0 = (_f32(1.0) - ratio) / (_f32(1.0) + ratio)
                    r0 = r0 * r0
                    om = _f32(1.0) - cos_t
                    refl_p = r0 + (_f32(1.0) - r0) * (np.float32(np.nan) if om < 0 else ((om * om) * (om * om)) * om)
                    if ratio * sin_t > 1 or refl_p > draws[2]:
                        k2 = _f32(2.0) * _dot3(ud, nf)
                        scv = [ud[k] - nf[k] * k2 for k in range(3)]
                        new_pos = [hpos[k] + nf[k] * _f32(1e-4) for k in range(3)]
                    else:
                        perp = [(ud[k] + nf[k] * cos_t) * ratio for k in range(3)]
                        kk = _f32(1.0) - _dot3(perp, perp)
                        par = -np.sqrt(max(kk, _f32(0.0)))
                        scv = [perp[k] + nf[k] * par for k in range(3)]
                        new_pos = [hpos[k] - nf[k] * _f32(1e-4) for k in range(3)]
                    thr = [thr[k] * albedo[k] for k in range(3)]
                    new_dir = _normalize3(scv)
                assert np.array_equal(rays["pos"][i].view("<u4"), np.array(new_pos, "<f4").view("<u4")), (flip, i)
                assert np.array_equal(rays["dir"][i].view("<u4"), np.array(new_dir, "<f4").view("<u4")), (flip, i)
                assert rays["last_inst"][i] == rays["last_prim"][i] == 0x7FFFFFFF  # origins are offset instead
                for k in range(3):
                    x = (_f32(0.0) + thr[k] * _f32(10.0)) * _f32(1000.0)
                    assert int(got4[4 * i + k]) == int(np.uint32(x)), (flip, i, k)
                assert gotc[i] == 1


def test_instance_matrices_against_a_numpy_restatement_of_the_shader_text():
    # Transform.matrix() / matrix_inverse() (shaders/common.slang:41-111) a second time: 4x4
    # products summed left to right in numpy float32 with the C library's sinf / cosf, observed
    # through a rotated, scaled, translated rect — rays moved to object space with M^-1, the
    # triangle test there, front_face from normalize(M n) — against the C oracle's hits
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    sin = lambda x: _f32(libm.sinf(float(x)))
    cos = lambda x: _f32(libm.cosf(float(x)))

    def mul(a, b):
        o = [[None] * 4 for _ in range(4)]
        for i in range(4):
            for j in range(4):
                acc = a[i][0] * b[0][j]
                for k in (1, 2, 3):
                    acc = acc + a[i][k] * b[k][j]
                o[i][j] = acc
        return o

    def ident():
        return [[_f32(1.0 if i == j else 0.0) for j in range(4)] for i in range(4)]

    scale, rot, tr = (2.0, 0.75, 1.5), (0.4, -1.1, 2.3), (0.3, -0.2, 3.0)
    sc_, ro, tl = ([_f32(v) for v in x] for x in (scale, rot, tr))
    rx, ry, rz = ident(), ident(), ident()
    rx[1][1], rx[1][2], rx[2][1], rx[2][2] = cos(ro[0]), -sin(ro[0]), sin(ro[0]), cos(ro[0])
    ry[0][0], ry[0][2], ry[2][0], ry[2][2] = cos(ro[1]), sin(ro[1]), -sin(ro[1]), cos(ro[1])
    rz[0][0], rz[0][1], rz[1][0], rz[1][1] = cos(ro[2]), -sin(ro[2]), sin(ro[2]), cos(ro[2])
    R = mul(rx, mul(ry, rz))
    S, T, Si, Ti = ident(), ident(), ident(), ident()
    for k in range(3):
        S[k][k] = sc_[k]
        Si[k][k] = _f32(1.0) / sc_[k]
        T[k][3] = tl[k]
        Ti[k][3] = -tl[k]
    M = mul(T, mul(R, S))
    Rt = [[R[j][i] for j in range(4)] for i in range(4)]
    Mi = mul(Si, mul(Rt, Ti))
    xv = lambda m, p: [_fma(m[r][2], p[2], _fma(m[r][1], p[1], m[r][0] * p[0])) for r in range(3)]
    xp = lambda m, p: [xv(m, p)[r] + m[r][3] for r in range(3)]

    s = host.Scene()
    rect = s.load_rect()
    mat = s.add_material(host.default_material())
    s.spawn(host.make_transform(scale=scale, rotation=rot, translation=tr), mat, rect)
    arrays = s.bind()
    pos4, nrm4, faces, _ = s.mesh(rect)
    s.close()
    rng = np.random.default_rng(31)
    n = 300
    uv = rng.uniform(-0.8, 0.8, (n, 2))
    uv = uv[(np.abs(np.abs(uv) - 0.5) > 0.04).all(axis=1)]  # clearly inside or clearly outside the rect
    n = len(uv)
    target = np.array([[float(x) for x in xp(M, [_f32(a), _f32(b), _f32(0.0)])] for a, b in uv])
    origin = rng.uniform(-2.0, 2.0, (n, 3)) + np.array([0.0, 0.0, 0.5])
    d = target - origin
    rays = np.zeros(n, ffi.Ray)
    rays["pos"] = origin.astype("<f4")
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rays["last_inst"] = rays["last_prim"] = ffi.NO_PRIM
    got = po.Scene(arrays).trace_closest(rays)
    hits = 0
    with np.errstate(all="ignore"):
        for i in range(n):
            wpos = [_f32(v) for v in rays["pos"][i]]
            wdir = [_f32(v) for v in rays["dir"][i]]
            opos, odir = xp(Mi, wpos), xv(Mi, wdir)  # the direction is not renormalised: t stays in world units
            best, t = None, np.float32(3.4028234663852886e38)
            for p, face in enumerate(faces):
                p0, p1, p2 = ([_f32(v) for v in pos4[face[k]][:3]] for k in range(3))
                e1 = [p1[k] - p0[k] for k in range(3)]
                e2 = [p2[k] - p0[k] for k in range(3)]
                q = _cross3(odir, e2)
                alpha = _dot3(e1, q)
                if -_f32(10e-8) < alpha < _f32(10e-8):
                    continue
                f = _f32(1.0) / alpha
                sv = [opos[k] - p0[k] for k in range(3)]
                u = f * _dot3(sv, q)
                if u < 0:
                    continue
                r = _cross3(sv, e1)
                v = f * _dot3(odir, r)
                if v < 0 or u + v > 1:
                    continue
                t2 = f * _dot3(e2, r)
                if t2 > t or t2 < 0:
                    continue
                t, best = t2, (p, u, v)
            inside = bool((np.abs(uv[i]) < 0.5).all())
            assert (best is not None) == inside, i
            if best is None:
                assert got["instance_id"][i] == 0xFFFFFFFF, i
                continue
            hits += 1
            p, u, v = best
            w = _f32(1.0) - u - v
            n0, n1, n2 = ([_f32(x) for x in nrm4[faces[p][k]][:3]] for k in range(3))
            nobj = [_fma(n2[k], v, _fma(n1[k], u, n0[k] * w)) for k in range(3)]
            normal = _normalize3(xv(M, nobj))
            assert (int(got["instance_id"][i]), int(got["triangle_id"][i])) == (0, p), i
            assert got["t"][i].view("<u4") == np.float32(t).view("<u4"), i
            assert got["front_face"][i] == (1 if _dot3(normal, wdir) < 0 else 0), i
    assert 40 < hits < n - 40


def test_light_sampling_and_connect_rays_against_a_numpy_restatement():
    # The connect stage with its documented repairs (light face drawn uniformly from the light's
    # BLAS root, point folded into the triangle, any-hit visibility up to the light, pdf from the
    # transformed edges; archive/light.slang:14-81, archive/shadow.slang:63-139) and the Lambertian
    # kernel's combination of the light sample with the scatter (archive/material.slang:17-139), a
    # second time in numpy float32: the shadow ray (origin, direction, tmax, excluded triangle) and
    # the sample's sums.  A Lambertian floor seen from its front, an emissive rect off-screen above
    # it, nothing in between; translation and axis scale only, so the matrices are exact.
    import ctypes as C
    W, H = 12, 8
    src = host.make_sample_sources(W, H, 9)
    cam = host.default_camera(W, H)
    colour = (0.7, 0.5, 0.3, 1.0)
    emit = (20.0, 15.0, 10.0, 0.0)
    light_at = (0.0, 6.0, 1.0)
    s = host.Scene()
    rect = s.load_rect()
    floor_m = s.add_material(host.make_material(colour=colour))
    light_m = s.add_material(host.make_material(colour=(1.0, 1.0, 1.0, 0.0), emissive=emit))
    s.spawn(host.make_transform(scale=(16.0, 16.0, -1.0), translation=(0.0, 0.0, 3.0)), floor_m, rect)
    s.spawn(host.make_transform(translation=light_at), light_m, rect)
    arrays = s.bind()
    pos4, nrm4, faces, _ = s.mesh(rect)
    s.close()
    assert list(arrays["light_sources"]) == [1]
    prm = po.default_params(max_bounces=2, seed=77, shade_mode=1, nee=1)
    sc = po.Scene(arrays)
    got4, gotc, _ = sc.render(cam, src, W, H, prm, 0, 1)
    crays = np.zeros(len(src), ffi.Ray)
    ctmax = np.zeros(len(src), "<f4")
    po.lib().oracle_connect_rays(sc._h, cam.ctypes.data, src.ctypes.data, len(src), W, H, C.byref(prm), 0, 0,
                                 crays.ctypes.data, ctmax.ctypes.data)
    albedo = [_f32(v) for v in colour[:3]]
    lrad = [_f32(v) for v in emit[:3]]
    lt = [_f32(v) for v in light_at]
    checked = 0
    with np.errstate(all="ignore"):
        for i in range(0, len(src), 3):
            draws, _ = po.rng_sequence(po.rng_init(prm.seed, i, 0), 12)
            draws = [_f32(v) for v in draws]
            wdir = _camera_dir(cam, src[i], W, H, draws[0], draws[1])
            hpos, normal, ff = _rect_hit(wdir, -1.0, pos4, nrm4, faces)
            assert ff
            nf = normal
            orig = [hpos[k] + nf[k] * _f32(1e-3) for k in range(3)]
            r1, r2, a, b = draws[2:6]
            li = min(int(np.floor(r1 * _f32(1.0))), 0)
            fi = min(int(np.floor(r2 * _f32(2.0))), 1)
            face = faces[fi]
            p0, p1, p2 = ([_f32(v) for v in pos4[face[k]][:3]] for k in range(3))
            e1 = [p1[k] - p0[k] for k in range(3)]
            e2 = [p2[k] - p0[k] for k in range(3)]
            if a + b > 1:
                a, b = _f32(1.0) - a, _f32(1.0) - b
            pobj = [(p0[k] + e1[k] * a) + e2[k] * b for k in range(3)]
            spos = [pobj[k] + lt[k] for k in range(3)]  # M = a pure translation
            cr = _cross3(e1, e2)
            area = _f32(0.5) * np.sqrt(_dot3(cr, cr))
            sprob = (_f32(1.0) / area) * (_f32(1.0) / _f32(2.0))
            ln = _normalize3(cr)
            d = [spos[k] - orig[k] for k in range(3)]
            shadow_t = np.sqrt(_dot3(d, d))
            scat = _normalize3(d)
            assert _dot3(scat, nf) > 0  # the light is on the floor's front side
            assert np.array_equal(crays["pos"][i].view("<u4"), np.array(orig, "<f4").view("<u4")), i
            assert np.array_equal(crays["dir"][i].view("<u4"), np.array(scat, "<f4").view("<u4")), i
            assert (int(crays["last_inst"][i]), int(crays["last_prim"][i])) == (1, fi), i
            assert ctmax[i].view("<u4") == np.float32(shadow_t).view("<u4"), i
            prob = sprob * _dot3(d, d) / abs(_dot3(ln, [-x for x in scat]))  # nothing occludes
            # the Lambertian kernel
            wo = [-x for x in wdir]
            sa, sb, scc = draws[6], draws[7], draws[8]
            unit = _normalize3([_f32(1.0) - _f32(2.0) * sa, _f32(1.0) - _f32(2.0) * sb, _f32(1.0) - _f32(2.0) * scc])
            scv = [nf[k] + unit[k] for k in range(3)]
            wi = _normalize3(nf if np.sqrt(_dot3(scv, scv)) < _f32(1e-6) else scv)
            lpdf = lambda w_: max(_f32(0.0), _dot3(w_, nf)) / _f32(3.1415)
            brdf = lambda w_, k: _f32(0.0) if (_dot3(nf, w_) <= 0 or _dot3(nf, wo) <= 0) else albedo[k] / _f32(3.1415)
            spdf = lpdf(wi)
            assert spdf > 0
            # where the bounce ray goes: leave the rare ones that reach the light's plane near the light alone
            new_pos = [float(hpos[k] + nf[k] * _f32(1e-3)) for k in range(3)]
            tt = (1.0 - new_pos[2]) / float(wi[2]) if wi[2] != 0 else -1.0
            if tt > 0:
                hx, hy = new_pos[0] + tt * float(wi[0]), new_pos[1] + tt * float(wi[1])
                if abs(hx) < 0.6 and abs(hy - 6.0) < 0.6:
                    continue
            for k in range(3):
                rad, thr = _f32(0.0), _f32(1.0)
                if prob > 0:
                    wgt = (_f32(1.0) * prob) / (_f32(1.0) * prob + _f32(1.0) * lpdf(scat))
                    c = thr * brdf(scat, k)
                    c = c * _dot3(scat, nf)
                    c = c * wgt
                    c = c / (_f32(1.0) * prob)
                    rad = rad + lrad[k] * c
                x = thr * brdf(wi, k)
                x = x * _dot3(wi, nf)
                x = x * ((_f32(1.0) * spdf) / (_f32(1.0) * spdf + _f32(1.0) * _f32(0.0)))
                thr = x / (_f32(1.0) * spdf)
                rad = rad + thr * _f32(10.0)  # the bounce ray leaves for the sky
                assert int(got4[4 * i + k]) == int(np.uint32(rad * _f32(1000.0))), (i, k)
            checked += 1
    assert checked > 20


def test_asset_facts_teapot_takes_the_computed_normal_path():
    s = host.Scene()
    mid = s.load_model_file(os.path.join(GOLDEN, "teapot.mirtmesh"))
    pos, nrm, faces, nodes = s.mesh(mid)
    assert len(pos) == 3644 and len(faces) == 6320
    ln = np.linalg.norm(nrm[:, :3], axis=1)
    assert np.allclose(ln, 1.0, atol=1e-5) and np.all(nrm[:, 3] == 0.0)
    # per-axis normalisation maps the largest positive centred coordinate to 1 (src/mesh.rs:322-332)
    assert np.array_equal(pos[:, :3].max(0), np.ones(3, "<f4")) and np.all(pos[:, 3] == 1.0)
    leaf = nodes[nodes["is_leaf"] == 1]
    assert int((leaf["end"] - leaf["start"]).sum()) == 6320


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
def test_obj_fixtures_are_what_the_loader_reads_from_the_reference_assets(tmp_path):
    L = ffi.host_lib()
    for name in ("suzanne", "teapot"):
        out = tmp_path / f"{name}.mirtmesh"
        assert L.mirth_obj_to_model_file(f"{REFERENCE}/assets/{name}.obj".encode(), str(out).encode()) == 0
        assert out.read_bytes() == open(os.path.join(GOLDEN, f"{name}.mirtmesh"), "rb").read()


def test_obj_parser_forms_quads_polygons_negative_indices_and_first_model_only():
    text = b"""# comment
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 1.5 0
vn 0 0 1
vt 0 0
f 1/1/1 2/1/1 3/1/1 4/1/1
f -5//1 -4//1 -3//1
f 1//1 2//1 3//1 4//1 5//1
p 1
l 1 2
o second
v 9 9 9
f 1//1 2//1 6//1
"""
    n_pos, n_nrm, n_tri = (ffi.C.c_uint32(), ffi.C.c_uint32(), ffi.C.c_uint32())
    assert ffi.host_lib().mirth_parse_obj_counts(text, len(text), n_pos, n_nrm, n_tri) == 0
    # quad -> 2, triangle -> 1, pentagon -> 3 (fan); the second object is ignored;
    # positions are the 5 used vertices, normals the 1 used normal
    assert (n_pos.value, n_nrm.value, n_tri.value) == (5, 1, 6)
    bad = b"v 0 0 0\nf 1 2 3\n"
    assert ffi.host_lib().mirth_parse_obj_counts(bad, len(bad), n_pos, n_nrm, n_tri) != 0
    empty = b"v 0 0 0\n"
    assert ffi.host_lib().mirth_parse_obj_counts(empty, len(empty), n_pos, n_nrm, n_tri) != 0
    # a file whose last line is a face ending in a digit, with no newline and no NUL behind it:
    # the reader is handed (pointer, length) into the middle of a larger buffer whose next bytes
    # are digits, which an unbounded integer parse would swallow
    tail = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"
    arena = ffi.C.create_string_buffer(tail + b"99999\n", len(tail) + 6)
    assert ffi.host_lib().mirth_parse_obj_counts(arena, len(tail), n_pos, n_nrm, n_tri) == 0
    assert (n_pos.value, n_nrm.value, n_tri.value) == (3, 0, 1)


def test_binder_packing_rules(config_scenes):
    arrays, info = config_scenes(3, dragon_n_u=64, dragon_n_v=16)
    # simple_scene: cube, dragon, rect meshes; 4 rect walls + dragon (src/scenes.rs:136-193)
    assert len(arrays["geometry_offsets"]) == 3 and len(arrays["instances"]) == 5
    # materials are de-duplicated in first-seen order: gray, red, green, glass
    assert len(arrays["materials"]) == 4
    assert arrays["instances"]["transform"].tolist() == [0, 1, 2, 3, 4]
    assert arrays["light_sources"].tolist() == [0xFFFFFFFF]  # no emissive instance (src/binder.rs:251-255)
    assert sorted(arrays["tlas_to_instances"].tolist()) == [0, 1, 2, 3, 4]
    t = arrays["tlas_nodes"]
    assert t[0]["right"] == 0 and all(t[t["is_leaf"] == 1]["end"] - t[t["is_leaf"] == 1]["start"] == 1)


def test_sample_sources_cover_every_pixel_once():
    for w, h in ((256, 256), (200, 120)):
        src = host.make_sample_sources(w, h, 42)
        lin = src["out_pos"][:, 0].astype(np.int64) + src["out_pos"][:, 1].astype(np.int64) * w
        assert np.array_equal(np.sort(lin), np.arange(w * h))
        assert np.allclose(src["screen_pos"][:, 0], src["out_pos"][:, 0] / np.float32(w))
        again = host.make_sample_sources(w, h, 42)
        assert np.array_equal(src, again)


# ---------------------------------------------------------------- traversal
def test_bvh_traversal_equals_brute_force(config_scenes, small_knot_scene):
    for arrays in (config_scenes(1)[0], config_scenes(2)[0], small_knot_scene):
        sc = po.Scene(arrays)
        rays = random_rays(3000, 11)
        a = sc.trace_closest(rays)
        b = sc.trace_brute(rays)
        assert np.array_equal(a["instance_id"], b["instance_id"])
        assert np.array_equal(a["triangle_id"], b["triangle_id"])
        assert np.array_equal(a["t"].view("<u4"), b["t"].view("<u4"))
        assert (a["instance_id"] != ffi.NO_HIT).sum() > 300


def test_last_primitive_is_excluded(config_scenes):
    arrays, _ = config_scenes(1)
    sc = po.Scene(arrays)
    rays = random_rays(2000, 5)
    first = sc.trace_closest(rays)
    hit = first["instance_id"] != ffi.NO_HIT
    again = rays.copy()
    again["last_inst"] = np.where(hit, first["instance_id"], ffi.NO_PRIM)
    again["last_prim"] = np.where(hit, first["triangle_id"], ffi.NO_PRIM)
    second = sc.trace_closest(again)
    same = (second["instance_id"] == first["instance_id"]) & (second["triangle_id"] == first["triangle_id"]) & hit
    assert not same.any()
    assert np.all(second["t"][hit] >= first["t"][hit])


def test_any_hit_agrees_with_closest_hit_on_occlusion(config_scenes):
    arrays, _ = config_scenes(2)
    sc = po.Scene(arrays)
    rays = random_rays(4000, 9)
    closest = sc.trace_closest(rays)
    tmax = np.full(len(rays), 3.0, "<f4")
    anyh = sc.trace_any(rays, tmax)
    occluded = anyh["instance_id"] != ffi.NO_HIT
    assert np.array_equal(occluded, closest["t"] <= 3.0)
    assert np.all(anyh["t"][occluded] >= closest["t"][occluded])


def test_golden_ray_sets(config_scenes):
    for cfg in (1, 2):
        arrays, _ = config_scenes(cfg)
        sc = po.Scene(arrays)
        for bounce in (0, 1):
            rays = np.ascontiguousarray(GOLD[f"cfg{cfg}_bounce{bounce}_rays"]).view(ffi.Ray).reshape(-1)
            want = np.ascontiguousarray(GOLD[f"cfg{cfg}_bounce{bounce}_hits"]).view(ffi.HitId).reshape(-1)
            assert hits_equal(sc.trace_closest(rays), want)


# ------------------------------------------------------------------ render
def _render_small(config_scenes, cfg, W, H, spp, mode):
    arrays, info = config_scenes(cfg)
    sc = po.Scene(arrays)
    src = host.make_sample_sources(W, H, info.seed)
    cam = host.default_camera(W, H)
    prm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode)
    return sc, src, cam, prm


@pytest.mark.parametrize("cfg,W,H,spp", [(1, 64, 64, 2), (2, 64, 36, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_render_sums(config_scenes, cfg, W, H, spp, mode):
    sc, src, cam, prm = _render_small(config_scenes, cfg, W, H, spp, mode)
    s4, cnt, st = sc.render(cam, src, W, H, prm, 0, spp)
    assert np.array_equal(s4, GOLD[f"cfg{cfg}_mode{mode}_sum4"])
    assert np.array_equal(cnt, GOLD[f"cfg{cfg}_mode{mode}_count"])
    assert [st.extend_rays, st.connect_rays, st.shade_events] == GOLD[f"cfg{cfg}_mode{mode}_rays"].tolist()
    assert np.all(cnt == spp)


@pytest.mark.parametrize("mode", [0, 1])
def test_wavefront_simulation_equals_per_sample_loop(config_scenes, mode):
    # queue.slang LIFO queues + frame order sample -> extend -> shade with few slots:
    # the stream is keyed per sample, so slot assignment and queue order cannot matter
    sc, src, cam, prm = _render_small(config_scenes, 1, 48, 48, 2, mode)
    a4, ac, ast = sc.render(cam, src, 48, 48, prm, 0, 2)
    for slots in (97, 48 * 48):
        b4, bc, bst, frames = sc.render_wavefront(cam, src, 48, 48, prm, 0, 2, slots)
        assert np.array_equal(a4, b4) and np.array_equal(ac, bc)
        assert (ast.extend_rays, ast.shade_events, ast.connect_rays) == (bst.extend_rays, bst.shade_events, bst.connect_rays)
        assert frames > 2


def test_sample_ranges_add_up(config_scenes):
    sc, src, cam, prm = _render_small(config_scenes, 1, 40, 40, 3, 0)
    full4, fullc, _ = sc.render(cam, src, 40, 40, prm, 0, 3)
    acc4 = np.zeros_like(full4)
    accc = np.zeros_like(fullc)
    for rank in range(2):
        first, n = pathtracer.partition_samples(3, 2, rank)
        sc.render(cam, src, 40, 40, prm, first, n, sum4=acc4, count=accc)
    assert np.array_equal(acc4, full4) and np.array_equal(accc, fullc)


def test_partition_samples_is_a_partition():
    for spp in (1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            parts = [pathtracer.partition_samples(spp, world, r) for r in range(world)]
            covered = []
            for first, n in parts:
                covered += list(range(first, first + n))
            assert covered == list(range(spp))


def test_white_furnace_every_sample_is_the_sky(config_scenes):
    # a camera that looks away from everything: every path misses, rad = throughput * sky = 10
    arrays, info = config_scenes(1)
    sc = po.Scene(arrays)
    W = H = 32
    src = host.make_sample_sources(W, H, 1)
    cam = host.default_camera(W, H)
    cam["forward"][0] = (0.0, 0.0, -1.0)
    prm = po.default_params(max_bounces=4)
    s4, cnt, st = sc.render(cam, src, W, H, prm, 0, 3)
    assert np.all(cnt == 3)
    assert np.all(s4.reshape(-1, 4)[:, :3] == 30000) and np.all(s4.reshape(-1, 4)[:, 3] == 0)
    assert st.shade_events == 0 and st.extend_rays == 3 * W * H
    rgb = po.resolve(s4, cnt, src, W, H, prm)
    assert len(np.unique(rgb)) == 1


def test_oracle_rejects_inconsistent_scene_buffers(config_scenes):
    arrays, _ = config_scenes(1)
    bad = {k: v.copy() for k, v in arrays.items()}
    bad["instances"]["geometry"][0] = 99
    with pytest.raises(ValueError):
        po.Scene(bad)
    bad = {k: v.copy() for k, v in arrays.items()}
    bad["indices"]["v"][0, 0] = 10 ** 6
    with pytest.raises(ValueError):
        po.Scene(bad)


def test_procedural_meshes_use_their_own_trigonometry_and_have_pinned_bits():
    # the stand-in geometry must not depend on the host's libm: sin / cos come from IEEE +, -, *
    # on doubles (checked here against numpy to an ulp), and the meshes built from them hash
    # to the values computed when the generator was written
    import ctypes as C
    import hashlib
    L = ffi.host_lib()
    s_out, c_out = C.c_double(), C.c_double()
    worst = 0.0
    for x in np.concatenate([np.linspace(-50.0, 900.0, 4001), [0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, 1e-300]]):
        L.mirth_det_sincos(float(x), C.byref(s_out), C.byref(c_out))
        worst = max(worst, abs(s_out.value - np.sin(x)), abs(c_out.value - np.cos(x)))
    assert worst < 2.3e-16
    s = host.Scene()
    pos, _, faces, nodes = s.mesh(s.load_knot(96, 24, 7))
    assert (len(pos), len(faces), len(nodes)) == (2304, 4608, 3287)
    assert hashlib.sha256(pos.tobytes()).hexdigest()[:32] == "6c9e645ed123bd2e551fd73e77e7e200"
    # suzanne with every triangle split in four, twice: 968 * 16 triangles, V + E new vertices per level
    pos, nrm, faces, nodes = s.mesh(s.load_subdivided("suzanne", 2))
    assert (len(pos), len(faces), len(nodes)) == (7824, 15488, 11255)
    assert hashlib.sha256(pos.tobytes()).hexdigest()[:32] == "d2f94e268460238522170f72d730122b"
    assert np.all(np.isfinite(nrm)) and pos[:, :3].max() <= 1.0 + 1e-6  # normalised like any loaded mesh (+ side to 1)
    s.close()
    # configuration 3 with the subdivided asset standing in for the dragon
    arrays, info = host.build_config_scene(3, standin="suzanne", standin_levels=2)
    assert info.procedural_dragon == 2 and len(arrays["indices"]) == 12 + 15488 + 2
    arrays, info = host.build_config_scene(3, dragon_n_u=64, dragon_n_v=16)
    assert info.procedural_dragon == 1 and len(arrays["indices"]) == 12 + 2 * 64 * 16 + 2


def test_tlas_of_the_host_library_against_the_python_restatement(config_scenes):
    # TLAS::new (src/tlas.rs:61-158): eight corners of every instance's BLAS root box through
    # translate * (Rx * (Ry * Rz)) * scale with glam's column-major products, then the same
    # builder with leaves of one — rebuilt from the flat buffers by an independent restatement
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    sin = lambda x: np.float32(libm.sinf(float(x)))
    cos = lambda x: np.float32(libm.cosf(float(x)))
    for cfg, kw in ((2, {}), (3, dict(dragon_n_u=48, dragon_n_v=12)), (4, dict(dragon_n_u=48, dragon_n_v=12))):
        arrays, _ = config_scenes(cfg, **kw)
        inst = arrays["instances"]
        roots = arrays["blas_nodes"][arrays["geometry_offsets"]["blas_node"]]
        geom_aabbs = [(r["lb"][:3].copy(), r["ub"][:3].copy()) for r in roots]
        transforms = [(t["scale"][:3], t["rotation"][:3], t["translation"][:3]) for t in arrays["transforms"]]
        want_nodes, want_ids = arrays["tlas_nodes"], arrays["tlas_to_instances"]
        nodes, ids = host_ref.build_tlas(geom_aabbs, transforms, [(int(i["transform"]), int(i["geometry"])) for i in inst],
                                         sin, cos)
        assert ids == want_ids.tolist()
        got = host_ref.nodes_as_records(nodes)
        assert len(got) == len(want_nodes)
        for g, w in zip(got, want_nodes):
            assert g == (tuple(float(x) for x in w["lb"][:3]), tuple(float(x) for x in w["ub"][:3]), int(w["left"]),
                         int(w["right"]), int(w["is_leaf"]), int(w["start"]), int(w["end"]))
