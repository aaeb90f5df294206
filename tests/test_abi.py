"""The C ABI: both libraries load and export every symbol their headers declare
(no GPU needed: nothing here launches a kernel), struct sizes match the
reference layouts, and the N>1 reduction path works over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from raytracer_amd import ffi


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_device_library_exports_every_declared_symbol():
    names = _declared("mirt.h", "mirt_")
    assert len(names) >= 25
    L = ffi.device_lib()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(ffi.DEVICE_SYMBOLS) == names


def test_host_library_exports_every_declared_symbol():
    names = _declared("mirt_host.h", "mirth_")
    L = ffi.host_lib()
    for n in names:
        assert hasattr(L, n), n


def test_static_archive_is_built():
    assert os.path.exists(os.path.join(ROOT, "raytracer_amd", "lib", "libmirt.a"))


def test_struct_sizes_match_the_reference_layouts():
    # compile a tiny C program against the header and print sizeof()
    src = textwrap.dedent("""
        #include <stdio.h>
        #include "mirt.h"
        int main(void) {
          printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(MirtVertex), sizeof(MirtIndex),
                 sizeof(MirtBVHNode), sizeof(MirtGeometryOffsets), sizeof(MirtInstance), sizeof(MirtTransform),
                 sizeof(MirtMaterial), sizeof(MirtCamera), sizeof(MirtSampleSource), sizeof(MirtRay), sizeof(MirtHitId),
                 sizeof(MirtParams), sizeof(MirtSceneDesc));
          return 0;
        }""")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes[:11] == [48, 16, 64, 12, 12, 48, 64, 64, 24, 32, 16]
    assert sizes[11] == C.sizeof(ffi.Params) and sizes[12] == C.sizeof(ffi.SceneDesc)


def test_plain_c_example_builds_against_both_libraries(tmp_path):
    # examples/render_c.c uses nothing but the two headers and links both libraries
    exe = str(tmp_path / "render_c")
    lib = os.path.join(ROOT, "raytracer_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "render_c.c"), "-L", lib, "-lmirt", "-lmirt_host",
                           "-Wl,-rpath," + lib, "-o", exe])
    # without arguments it prints its usage and touches no device
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_status_strings_and_default_params_need_no_device():
    L = ffi.device_lib()
    assert L.mirt_status_string(0) == b"ok"
    assert b"scene" in L.mirt_status_string(-6)
    p = ffi.Params()
    L.mirt_default_params(C.byref(p))
    # the reference's literals: sample.slang:71, ray_extend.slang:36, sample.slang:45-47,54
    assert (p.max_bounces, p.sky_radiance, p.fixed_point_scale) == (128, 10.0, 1000.0)
    assert np.float32(p.exposure) == np.exp2(np.float32(-2.5))


def test_no_exception_leaves_the_host_library():
    """A request the host cannot allocate (a knot mesh of 2^40 vertices) comes back as a status and a
    message — every entry point of both libraries is a function-try-block, nothing is thrown across the C
    boundary (the caller is C, Rust or ctypes) — and the scene object stays usable."""
    H = ffi.host_lib()
    s = H.mirth_scene_new()
    try:
        assert H.mirth_load_mesh_knot(s, 1 << 20, 1 << 20, 1) == -1
        H.mirth_last_error.restype = C.c_char_p
        assert H.mirth_last_error(s) in (b"out of host memory",) or H.mirth_last_error(s).startswith(b"internal error")
        assert H.mirth_load_mesh_knot(s, 64, 8, 1) >= 0
    finally:
        H.mirth_scene_free(s)
    assert ffi.device_lib().mirt_status_string(-7) == b"internal error"


def test_two_rank_gloo_reduce_of_partial_accumulators():
    """Samples-per-pixel are partitioned over ranks, each rank accumulates its share,
    and one integer all-reduce combines them: bit-identical to a single-rank render.
    The ranks' device renders are stood in for by the CPU oracle here (no GPU)."""
    script = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        from raytracer_amd import host, pathtracer
        from raytracer_amd.distributed import allreduce_accumulators
        from oracle import pyoracle as po
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        arrays, info = host.build_config_scene(1)
        sc = po.Scene(arrays)
        W = H = 32; spp = 3
        src = host.make_sample_sources(W, H, info.seed); cam = host.default_camera(W, H)
        prm = po.default_params(max_bounces=4, seed=info.seed)
        first, n = pathtracer.partition_samples(spp, world, rank)
        s4, cnt, _ = sc.render(cam, src, W, H, prm, first, n)
        t4 = torch.from_numpy(s4.view(np.int32).copy()); tc = torch.from_numpy(cnt.view(np.int32).copy())
        allreduce_accumulators(t4, tc)
        full4, fullc, _ = sc.render(cam, src, W, H, prm, 0, spp)
        ok = np.array_equal(t4.numpy().view(np.uint32), full4) and np.array_equal(tc.numpy().view(np.uint32), fullc)
        # sums that overflow 32 bits only when the ranks' shares are added saturate, as they do on one GPU
        # and in mirt_group_reduce; the ones that fit are exact; counts wrap
        part = np.array([0xC0000000, 0x7FFFFFFF, 5, 0xFFFFFFFF, 0x80000000 if rank == 0 else 0x7FFFFFFF, 0, 1, 2], np.uint32)
        u4 = torch.from_numpy(part.view(np.int32).copy()); uc = torch.from_numpy(np.array([0xFFFFFFFF, 7], np.uint32).view(np.int32).copy())
        allreduce_accumulators(u4, uc)
        want = np.array([0xFFFFFFFF, 0xFFFFFFFE, 10, 0xFFFFFFFF, 0xFFFFFFFF, 0, 2, 4], np.uint32)
        ok = ok and np.array_equal(u4.numpy().view(np.uint32), want)
        ok = ok and np.array_equal(uc.numpy().view(np.uint32), np.array([0xFFFFFFFE, 14], np.uint32))
        print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
        """ % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.py")
        open(path, "w").write(script)
        env = dict(os.environ, OMP_NUM_THREADS="2")
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", "29617", path],
                             capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 2


def _bench(*argv, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=120, env=env)


def test_bench_starts_one_process_per_gpu_by_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh rank processes (before anything
    touches torch or HIP) with the rank environment torch.distributed.run would give them."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = _bench("--gpus", "2", "--dry-launch", env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
    assert sorted(l["RANK"] for l in lines) == ["0", "1"]
    assert all(l["WORLD_SIZE"] == "2" and l["LOCAL_RANK"] == l["RANK"] and l["MASTER_ADDR"] == "127.0.0.1" for l in lines)
    assert len({l["MASTER_PORT"] for l in lines}) == 1 and len({l["pid"] for l in lines}) == 2
    # under a launcher (WORLD_SIZE set) it is a rank itself and starts nothing
    out = _bench("--gpus", "2", "--dry-launch", env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2"))
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1 and lines[0]["RANK"] == "1"


def test_bench_launcher_reports_a_failed_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = _bench("--gpus", "2", "--dry-launch", env=dict(env, MIRT_BENCH_DRY_FAIL_RANK="1"))
    assert out.returncode != 0 and "ranks failed" in out.stderr
