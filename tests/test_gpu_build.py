"""GPU tests of the device BLAS builder (mirt_build_blas) against the host
builder, which restates BLAS::new / BVH::initialize of the reference
(src/blas.rs:62-78, src/bvh.rs:52-178) and is itself pinned by the hand-traced
cube and the pure-Python builder in tests/test_oracle_kat.py.

Bar: the reordered face array and the node array (boxes, children, skip links,
ranges) are compared BYTE FOR BYTE.
"""
import time

import numpy as np
import pytest

from raytracer_amd import ffi, host, pathtracer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    p = pathtracer.Pathtracer(0)
    yield p
    p.close()


def faces_of(indices):
    f = np.zeros((len(indices) // 3, 4), "<u4")
    f[:, :3] = np.asarray(indices, "<u4").reshape(-1, 3)
    return f


def positions4(xyz):
    p = np.ones((len(xyz), 4), "<f4")  # Mesh::positions carry w = 1
    p[:, :3] = xyz
    return p


def assert_same_blas(pt, name, xyz, indices):
    pos, faces0 = positions4(xyz), faces_of(indices)
    want_faces, want_nodes = host.build_blas(pos, faces0)
    faces, nodes, _ = pt.build_blas(pos, faces0)
    assert np.array_equal(faces, want_faces), name
    assert len(nodes) == len(want_nodes), (name, len(nodes), len(want_nodes))
    assert nodes.tobytes() == want_nodes.tobytes(), name


def soup(rng, n_vertices, n_faces):
    positions = rng.uniform(-1.0, 1.0, (n_vertices, 3)).astype("<f4")
    indices = rng.integers(0, n_vertices, (n_faces, 3)).astype("<u4")
    return positions, indices.reshape(-1)


@pytest.mark.parametrize("n_faces", [1, 2, 4, 5, 6, 7, 12, 63, 64, 65, 300, 2047, 2048, 2049, 5000, 16383, 16384, 16385, 70000, 200001])
def test_triangle_soups_of_every_size_class(pt, n_faces):
    # sizes around the leaf threshold (4), one wave (64), the switch to the 1024-thread
    # block shape (2048) and to nodes split by many blocks (16384, chunks of 4096)
    rng = np.random.default_rng(n_faces)
    positions, indices = soup(rng, max(3, n_faces // 2 + 3), n_faces)
    assert_same_blas(pt, f"soup{n_faces}", positions, indices)


def test_one_sided_partitions_and_equal_centroids(pt):
    rng = np.random.default_rng(5)
    # (a) many copies of one triangle: every centroid equal, the partition puts everything right,
    #     the node stays a leaf of thousands of triangles while its faces are still permuted
    positions = rng.uniform(-1.0, 1.0, (6, 3)).astype("<f4")
    indices = np.tile(np.array([0, 1, 2], "<u4"), 3000)
    indices[:9] = [3, 4, 5, 0, 1, 2, 1, 2, 3]
    assert_same_blas(pt, "copies", positions, indices)
    # the same with a root that many blocks split together: one-sided at the top, and a big
    # cluster of coincident triangles below a few ordinary splits
    indices = np.tile(np.array([0, 1, 2], "<u4"), 40000)
    assert_same_blas(pt, "copies_huge", positions, indices)
    positions, indices = soup(rng, 9000, 60000)
    tri = indices.reshape(-1, 3)
    tri[5000:45000:2] = tri[5000]
    assert_same_blas(pt, "clusters_huge", positions, tri.reshape(-1))
    # (b) clusters of coincident triangles among ordinary ones: deep, lopsided trees with
    #     bail-out leaves of mixed sizes
    positions, indices = soup(rng, 500, 6000)
    tri = indices.reshape(-1, 3)
    tri[1000:1800] = tri[1000]
    tri[3000:3005] = tri[3000]
    tri[4000:6000:2] = tri[4000]
    assert_same_blas(pt, "clusters", positions, tri.reshape(-1))
    # (c) grid-aligned geometry: many equal coordinates, ties in the axis choice, centroids on the split
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3), indexing="ij"), -1).reshape(-1, 3)
    positions = g.astype("<f4")
    indices = rng.integers(0, len(positions), (4000, 3)).astype("<u4").reshape(-1)
    assert_same_blas(pt, "grid", positions, indices)


def test_random_meshes_of_many_shapes(pt):
    # eighty seeded meshes: sizes from 5 to 60 000 faces, vertex clouds that are uniform, clustered,
    # flat, sorted along an axis or quantised to a coarse grid (many equal centroids and boxes),
    # with and without runs of duplicated faces
    rng = np.random.default_rng(2026)
    for case in range(80):
        n_faces = int(rng.choice([5, 9, 17, 33, 100, 511, 513, 3000, 20000, 60000]) + rng.integers(0, 7))
        n_vertices = max(4, int(n_faces * rng.choice([0.02, 0.5, 1.5])))
        kind = case % 5
        xyz = rng.uniform(-1.0, 1.0, (n_vertices, 3))
        if kind == 1:
            xyz = rng.normal(size=(n_vertices, 3)) * 0.01 + rng.integers(-2, 3, (n_vertices, 1))
        elif kind == 2:
            xyz[:, rng.integers(0, 3)] = 0.25
        elif kind == 3:
            xyz = xyz[np.argsort(xyz[:, 0])]
        elif kind == 4:
            xyz = np.round(xyz * 4.0) / 4.0
        indices = rng.integers(0, n_vertices, (n_faces, 3))
        if kind == 3:
            indices = np.sort(indices, axis=0)  # faces roughly ordered along x
        if case % 3 == 0 and n_faces > 20:
            a = int(rng.integers(0, n_faces // 2))
            indices[a:a + n_faces // 3] = indices[a]
        assert_same_blas(pt, f"random{case}", xyz.astype("<f4"), indices.astype("<u4").reshape(-1))


def test_signed_zero_bounds_follow_the_first_occurrence(pt):
    # coordinates that are +0 or -0 at random: the sign of a zero bound is that of the first
    # element in order, as the host's sequential fold keeps it — in leaves and in inner nodes
    rng = np.random.default_rng(9)
    n_vertices = 400
    xyz = rng.uniform(-1.0, 1.0, (n_vertices, 3)).astype("<f4")
    xyz[:, 2] = np.where(rng.random(n_vertices) < 0.5, 0.0, -0.0)
    xyz[::3, 0] = np.where(rng.random(len(xyz[::3])) < 0.5, 0.0, -0.0)
    indices = rng.integers(0, n_vertices, (3000, 3)).astype("<u4").reshape(-1)
    assert np.signbit(xyz[:, 2]).any() and not np.signbit(xyz[:, 2]).all()
    assert_same_blas(pt, "zeros", xyz, indices)


@pytest.mark.parametrize("cfg", [1, 2])
def test_scene_built_on_the_device_equals_the_host_build(pt, config_scenes, cfg):
    # the host model with the device builder plugged in (OBJ meshes, rect): identical flat buffers
    want, _ = config_scenes(cfg)
    s = host.Scene()
    s.set_blas_builder(pt)
    s.build_config(cfg)
    got = s.bind()
    s.close()
    for key in ("indices", "blas_nodes", "tlas_nodes", "vertices", "tlas_to_instances"):
        assert np.asarray(got[key]).tobytes() == np.asarray(want[key]).tobytes(), key


def test_dragon_class_mesh_at_full_size(pt, config_scenes):
    # config 3's 871 040-triangle mesh: byte-identical faces and nodes, and the time it takes
    want, _ = config_scenes(3)
    t0 = time.perf_counter()
    s = host.Scene()
    s.set_blas_builder(pt)
    s.build_config(3)
    got = s.bind()
    s.close()
    dt = time.perf_counter() - t0
    for key in ("indices", "blas_nodes", "tlas_nodes"):
        assert np.asarray(got[key]).tobytes() == np.asarray(want[key]).tobytes(), key
    print(f"config 3 scene with the device builder: {dt:.2f} s")


def test_bad_arguments_are_refused(pt):
    pos = np.zeros((3, 4), "<f4")
    faces = np.array([[0, 1, 7, 0]], "<u4")  # index out of range
    with pytest.raises(pathtracer.MirtError):
        pt.build_blas(pos, faces)
    faces, nodes, _ = pt.build_blas(pos, np.zeros((0, 4), "<u4"))  # the empty mesh keeps its root leaf
    assert len(nodes) == 1 and nodes[0]["is_leaf"] == 1 and len(faces) == 0
    # too little room for the nodes: refused, with the count that is needed, and the call works again after
    import ctypes as C
    rng = np.random.default_rng(3)
    xyz, indices = soup(rng, 60, 100)
    p4, f0 = positions4(xyz), faces_of(indices)
    want_faces, want_nodes = host.build_blas(p4, f0)
    small = np.zeros(3, ffi.BVHNode)
    f1 = f0.copy()
    n = C.c_uint32()
    rc = pt._L.mirt_build_blas(pt._h, p4.ctypes.data, len(p4), f1.ctypes.data, len(f1), small.ctypes.data, len(small),
                               C.byref(n), None)
    assert rc != 0 and n.value == len(want_nodes) and np.array_equal(f1, f0)  # the faces were left alone
    faces, nodes, _ = pt.build_blas(p4, f0)
    assert np.array_equal(faces, want_faces) and nodes.tobytes() == want_nodes.tobytes()


# ------------------------------------------------------------------ TLAS

def assert_same_tlas(pt, name, arrays):
    """mirt_build_tlas against the host model's TLAS::new (src/tlas.rs:61-158) for the same
    geometry boxes, transforms and instances: node array and instance ids byte for byte."""
    roots = arrays["blas_nodes"][arrays["geometry_offsets"]["blas_node"]]
    aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
    nodes, ids, ms = pt.build_tlas(aabbs, arrays["transforms"], arrays["instances"])
    assert np.array_equal(ids, arrays["tlas_to_instances"]), name
    assert len(nodes) == len(arrays["tlas_nodes"]), (name, len(nodes), len(arrays["tlas_nodes"]))
    assert nodes.tobytes() == arrays["tlas_nodes"].tobytes(), name
    return ms


def scattered_scene(n, seed):
    rng = np.random.default_rng(seed)
    s = host.Scene()
    meshes = [s.load_cube(), s.load_rect()]
    mat = s.add_material(host.make_material())
    for i in range(n):
        # clusters, coincident instances and degenerate scales next to ordinary ones
        centre = rng.integers(0, 7, 3) * 10.0 if i % 3 else np.zeros(3)
        t = host.make_transform(scale=rng.uniform(0.2, 3.0, 3) if i % 17 else (1.0, 0.0, 1.0),
                                rotation=rng.uniform(-4.0, 4.0, 3), translation=centre + rng.normal(0, 2.0, 3) * (i % 5 != 0))
        s.spawn(t, mat, meshes[i % 2])
    arrays = s.bind()
    s.close()
    return arrays


@pytest.mark.parametrize("n", [1, 2, 3, 16, 17, 65, 1000, 20001])
def test_tlas_of_scattered_instances(pt, n):
    # sizes around one-thread nodes (16), one wave (64) and nodes split by many blocks (16384)
    assert_same_tlas(pt, f"scattered{n}", scattered_scene(n, n))


def test_tlas_of_the_benchmark_scenes(pt, config_scenes):
    for cfg, kw in ((1, {}), (2, {}), (3, dict(dragon_n_u=48, dragon_n_v=12)), (4, dict(dragon_n_u=48, dragon_n_v=12))):
        arrays, _ = config_scenes(cfg, **kw)
        assert_same_tlas(pt, f"config{cfg}", arrays)
    # moved instances: rebuild on the device, upload, trace — same hits as the host-built scene
    arrays, _ = config_scenes(4, dragon_n_u=48, dragon_n_v=12)
    moved = {k: v.copy() for k, v in arrays.items()}
    rng = np.random.default_rng(5)
    moved["transforms"]["translation"][:, :3] += rng.normal(0, 0.3, (len(moved["transforms"]), 3)).astype("<f4")
    roots = moved["blas_nodes"][moved["geometry_offsets"]["blas_node"]]
    aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
    nodes, ids, _ = pt.build_tlas(aabbs, moved["transforms"], moved["instances"])
    moved["tlas_nodes"], moved["tlas_to_instances"] = nodes, ids
    pt.upload_scene(arrays)
    pt.update_instances(moved)
    from conftest import hits_equal, random_rays
    from oracle import pyoracle as po
    rays = random_rays(50000, 4, centre=(0.0, 0.0, 14.0), spread=10.0)
    assert hits_equal(pt.trace_closest(rays), po.Scene(moved).trace_closest(rays))


def test_transforms_only_update_stays_on_the_device(pt, config_scenes):
    """mirt_scene_update_transforms: new matrices, instance boxes, TLAS and its traversal layout without the
    host builder — the same hits as the host-bound moved scene through the oracle, frame after frame."""
    from conftest import hits_equal, random_rays
    from oracle import pyoracle as po
    arrays, _ = config_scenes(4, dragon_n_u=48, dragon_n_v=12)
    pt.upload_scene(arrays)
    rays = random_rays(50000, 4, centre=(0.0, 0.0, 14.0), spread=10.0)
    rng = np.random.default_rng(6)
    moved = {k: v.copy() for k, v in arrays.items()}
    for step in range(3):
        moved["transforms"]["translation"][:, :3] += rng.normal(0, 0.3, (len(moved["transforms"]), 3)).astype("<f4")
        moved["transforms"]["rotation"][:, :3] += rng.normal(0, 0.2, (len(moved["transforms"]), 3)).astype("<f4")
        ms = pt.update_transforms(moved["transforms"])
        assert ms > 0.0
        # what the host would have bound for these transforms (TLAS::new on the device is byte-identical to it)
        roots = moved["blas_nodes"][moved["geometry_offsets"]["blas_node"]]
        aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
        moved["tlas_nodes"], moved["tlas_to_instances"], _ = pt.build_tlas(aabbs, moved["transforms"], moved["instances"])
        want = po.Scene(moved).trace_closest(rays)
        assert hits_equal(pt.trace_closest(rays), want), step
    # and the same hits as the upload path for the same scene
    pt.update_instances(moved)
    assert hits_equal(pt.trace_closest(rays), want)
    with pytest.raises(pathtracer.MirtError):
        pt.update_transforms(moved["transforms"][:-1])


def test_transforms_only_update_refuses_a_tree_that_outgrew_its_room(pt):
    # ten coincident instances bind to a single leaf, so little room is reserved; moved apart they need 19 nodes
    s = host.Scene()
    cube = s.load_cube()
    mat = s.add_material(host.make_material())
    for _ in range(40):
        s.spawn(host.make_transform(), mat, cube)
    arrays = s.bind()
    s.close()
    assert len(arrays["tlas_nodes"]) == 1
    from conftest import hits_equal, random_rays
    rays = random_rays(2000, 9, centre=(0.0, 0.0, 0.0), spread=3.0)
    pt.upload_scene(arrays)
    before = pt.trace_closest(rays)
    apart = arrays["transforms"].copy()
    apart["translation"][:, 0] = np.arange(len(apart), dtype="<f4") * 5.0
    with pytest.raises(pathtracer.MirtError, match="outgrew"):
        pt.update_transforms(apart)
    assert hits_equal(pt.trace_closest(rays), before)   # the scene was left as it was


def test_transforms_only_update_refuses_a_tlas_that_leaves_instances_out(pt):
    """A caller-built TLAS may list fewer slots than there are instances (validation only asks that every slot
    names an instance).  The transforms-only update rebuilds a tree over every instance, so it refuses such a
    scene before any launch and leaves it as it was; the slot tables have room for every instance all the same."""
    from conftest import hits_equal, random_rays
    from oracle import pyoracle as po
    s = host.Scene()
    cube = s.load_cube()
    mat = s.add_material(host.make_material())
    for i in range(6):
        s.spawn(host.make_transform(translation=(3.0 * i, 0.0, 0.0)), mat, cube)
    arrays = s.bind()
    s.close()
    roots = arrays["blas_nodes"][arrays["geometry_offsets"]["blas_node"]]
    aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
    partial = {k: v.copy() for k, v in arrays.items()}
    partial["tlas_nodes"], partial["tlas_to_instances"], _ = pt.build_tlas(aabbs, arrays["transforms"], arrays["instances"][:4])
    assert len(partial["tlas_to_instances"]) == 4 and len(partial["instances"]) == 6
    rays = random_rays(4000, 11, centre=(7.5, 0.0, 0.0), spread=9.0)
    pt.upload_scene(partial)
    before = pt.trace_closest(rays)
    assert hits_equal(before, po.Scene(partial).trace_closest(rays))
    assert before["instance_id"][before["instance_id"] != ffi.NO_HIT].max() < 4   # the two instances left out are invisible
    moved = arrays["transforms"].copy()
    moved["translation"][:, 1] += 1.0
    with pytest.raises(pathtracer.MirtError, match="lists 4 slots for 6 instances"):
        pt.update_transforms(moved)
    assert hits_equal(pt.trace_closest(rays), before)
    # the full TLAS of the same instances takes the update
    pt.upload_scene(arrays)
    pt.update_transforms(moved)
    full = {k: v.copy() for k, v in arrays.items()}
    full["transforms"] = moved
    full["tlas_nodes"], full["tlas_to_instances"], _ = pt.build_tlas(aabbs, moved, arrays["instances"])
    assert hits_equal(pt.trace_closest(rays), po.Scene(full).trace_closest(rays))


def test_group_transforms_only_update_reaches_every_member():
    """mirt_group_scene_update_transforms: the device-side TLAS rebuild on every member of a group."""
    from conftest import hits_equal, random_rays
    from oracle import pyoracle as po
    s = host.Scene()
    cube = s.load_cube()
    mat = s.add_material(host.make_material())
    for i in range(9):
        s.spawn(host.make_transform(translation=(2.5 * (i % 3), 2.5 * (i // 3), 0.0)), mat, cube)
    arrays = s.bind()
    s.close()
    rays = random_rays(5000, 13, centre=(2.5, 2.5, 0.0), spread=7.0)
    moved = {k: v.copy() for k, v in arrays.items()}
    moved["transforms"]["translation"][:, 2] += np.linspace(-1.0, 1.0, len(moved["transforms"])).astype("<f4")
    moved["transforms"]["rotation"][:, 1] += 0.4
    with pathtracer.PathtracerGroup([0, 0]) as grp:
        grp.upload_scene(arrays)
        assert grp.update_transforms(moved["transforms"]) > 0.0
        roots = moved["blas_nodes"][moved["geometry_offsets"]["blas_node"]]
        aabbs = np.concatenate([roots["lb"][:, :3], roots["ub"][:, :3]], axis=1)
        moved["tlas_nodes"], moved["tlas_to_instances"], _ = grp.members[0].build_tlas(aabbs, moved["transforms"], moved["instances"])
        want = po.Scene(moved).trace_closest(rays)
        for m in grp.members:
            assert hits_equal(m.trace_closest(rays), want)
        with pytest.raises(pathtracer.MirtError, match="member 0"):
            grp.update_transforms(moved["transforms"][:-1])
