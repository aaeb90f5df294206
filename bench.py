#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (the headline metric's configuration, "dragon.obj 1080p, 8 bounces"):
the reference's simple_scene — open Cornell box + dragon — at 1920x1080 with
the reference's shuffled sample-source order, 8 bounces, sky 10.  dragon.obj is
absent from the reference checkout; unless tests/golden/dragon.{obj,mirtmesh}
exists, a deterministic procedural dragon-class mesh (871 040 triangles) stands
in, and `config.geometry` says so.

A step renders `--spp-per-step` samples for every pixel through the whole
pipeline (sample/new_ray -> extend -> shade, to completion) with all inputs
resident in HBM; by default that is configuration 3's own 256 samples per pixel, so
a step is the whole render BASELINE.json names (other configurations: 64).  With N > 1 every rank renders its own sample numbers
(samples-per-pixel are partitioned, scene replicated: weak scaling) and the
accumulators are combined by one RCCL all-reduce inside the timed region.

Prints ONE JSON line: Mrays/s (extend + connect rays actually traced, summed
over ranks, divided by the slowest rank's wall time), plus
  roofline      the dominant kernel (extend): algorithmic bytes per launch
                (reference-layout traffic model x visit counts counted on the
                device in an untimed replay of the same samples) / mean launch
                duration from HIP events on the kernels' stream
  cpu_baseline  the CPU oracle on the host cores, on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES = {1: "suzanne under a rect light",
          2: "open Cornell box + Lambertian/Metallic/Dielectric teapots",
          3: "simple_scene (open Cornell box + dragon)",
          4: "10x10x10 grid of dragon/teapot/suzanne/cube instances",
          5: "Cornell box with ceiling light + dielectric dragon, light sampling"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# what stands for dragon.obj (ConfigInfo.procedural_dragon)
GEOMETRY = {0: "dragon asset", 1: "procedural dragon-class knot, 871040 triangles (largest BVH leaf 2888)",
            2: "suzanne asset with every triangle split into four five times, jittered: 991232 triangles (largest BVH leaf 602)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samples per pixel one step renders; 0 (default): the configuration's own sample count when "
                         "that lies in 64..256 — config 3's 256 spp: a step is the whole render BASELINE.json names — "
                         "else 64")
    ap.add_argument("--paths-per-pixel", type=int, default=32,
                    help="paths in flight (slots of the wavefront state) per pixel")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--geometry", choices=["knot", "suzanne"], default="knot",
                    help="stand-in for the absent dragon asset: the procedural knot (871 040 triangles) or "
                         "suzanne with every triangle split into four five times (991 232 triangles)")
    ap.add_argument("--mode", choices=["render", "step"], default="render",
                    help="render: mirt_render_spp passes (the headline); step: mirt_step, one call per frame, at the "
                         "reference's operating point (512x512, one path slot per pixel, src/pathtracer.rs:34-38)")
    ap.add_argument("--frames-per-step", type=int, default=64, help="step mode: frames per bench step")
    ap.add_argument("--step-paths-per-pixel", type=int, default=1, help="step mode: path slots per pixel")
    ap.add_argument("--rng", choices=["per-sample", "per-slot"], default="per-sample",
                    help="step mode: per-sample keyed streams (the default everywhere) or the reference's persistent "
                         "stream per path slot (src/pathtracer_state.rs:98-110), which the Rust shim selects; the sample "
                         "stage then pairs slots with sample numbers in ascending slot order")
    ap.add_argument("--stage-kernels", action="store_true",
                    help="per-material shading with light sampling (config 5): a routing pass and one kernel per "
                         "material behind the connect kernel (MirtParams.stage_kernels) instead of one kernel")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--group", action="store_true",
                    help="with --gpus N: one process driving N GPUs through mirt_group_* (PathtracerGroup) instead "
                         "of N processes and an RCCL all-reduce")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise the multi-process launcher only: every rank prints its rank environment and exits "
                         "(no GPU, no torch)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second geometry's short run that the default one-GPU line carries as `secondary`")
    return ap.parse_args()


RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def device_of(local_rank):
    """The GPU a rank (or a group member) drives: its local rank, unless MIRT_BENCH_DEVICES lists ordinals
    ("0,0" rehearses two ranks on a one-GPU box; RCCL refuses two ranks on one device, so that rehearsal also
    sets MIRT_BENCH_BACKEND=gloo)."""
    listed = os.environ.get("MIRT_BENCH_DEVICES")
    return int(listed.split(",")[local_rank]) if listed else local_rank


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process — which has not imported torch
    or touched HIP, and never will — starts N fresh copies of itself, one per GPU, with the rank environment
    torch.distributed.run would give them, forwards rank 0's JSON line and returns non-zero if any rank did.
    Children are new processes (no exec of a process that has initialised the GPU)."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, LOCAL_WORLD_SIZE=str(n))
        # rank 0's stdout carries the line; the other ranks print nothing there, and every rank's stderr passes through
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves the others waiting in a collective: end them (by their own pids) instead of hanging
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def cpu_baseline(arrays, info, W, H, src, cam, target_seconds):
    """The CPU oracle (a port of the reference's kernels, OpenMP over samples) on
    this host's cores, on a bounded sample of the same workload: a pilot run sizes
    the timed run (sources x spp) to about `target_seconds` of CPU work."""
    from oracle import pyoracle as po
    sc = po.Scene(arrays)
    prm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=0)
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or (os.cpu_count() or 1)))
    n = min(len(src), 65536)
    t0 = time.perf_counter()
    _, _, st = sc.render(cam, src[:n].copy(), W, H, prm, 0, 1, threads=threads)
    dt = time.perf_counter() - t0
    pilot_rays = st.extend_rays + st.connect_rays
    rate = pilot_rays / max(dt, 1e-9)
    want_rays = target_seconds * rate
    rays_per_source_spp = pilot_rays / n
    n2 = int(min(len(src), max(n, want_rays / max(rays_per_source_spp, 1e-9))))
    spp = int(max(1, min(64, round(want_rays / max(rays_per_source_spp * n2, 1e-9)))))
    t0 = time.perf_counter()
    _, _, st = sc.render(cam, src[:n2].copy(), W, H, prm, 0, spp, threads=threads)
    dt = time.perf_counter() - t0
    rays = st.extend_rays + st.connect_rays
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"first {n2} of {len(src)} sample sources of the same workload, {spp} spp, {rays} rays in {dt:.2f} s"}


DEVICE_SOURCES = ["engine.h", "kernels.hip", "kernels.h", "layout.h", "device_util.h", "rt_math.h", "sample_stage.h",
                  "classic.hip", "api.hip"]


def device_sources_sha():
    """Hash of the sources the traversal and frame kernels are built from, comments and blank lines
    left out: a counter figure taken from profiles/ is only used when it was measured on this very code."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in DEVICE_SOURCES:
        with open(os.path.join(ROOT, "raytracer_amd", "csrc", "device", name), "r") as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        lines = [re.sub(r"//.*$", "", line).rstrip() for line in text.split("\n")]
        h.update("\n".join(line for line in lines if line.strip()).encode())
    return h.hexdigest()[:16]


def extend_roofline(pt, st, cst, elapsed, args, workload, n_paths, classic=False):
    """What bounds the extend kernel, from this run's own measurements.

    The walk is a per-lane chain of dependent small fetches and the scene sits in L2 +
    Infinity Cache, so the binding resource is the rate at which the memory pipeline serves
    per-lane gathers, not HBM streaming.
      achieved  bytes the lanes actually request per launch (32 B per node visit, 36 B per
                triangle fetched by a lane or by a whole-wave scan read from
                memory, 68 B per instance visit, 48 B per ray in and out) / the kernel's mean
                launch time (HIP events on its stream)
      peak      32 B x the lane-steps/s of dependent 32-byte gathers from a table that fits an
                XCD's L2 (mirt_bench_gather, measured now, untimed, same block shape): the most
                the L1 / texture-address path delivers to per-lane gathers, whatever the hit rates
    Triangle tests served from the LDS leaf cache move no memory and are listed apart.  Behind the
    L1 the second stage is the L2-miss path: fabric_frac = memory-side read requests per second
    (FETCH_SIZE / 64 B from profiles/traffic.json, used only when it was taken on these very
    sources) / the line rate of the same gather from a 192 MiB table (Infinity-Cache resident,
    2 % L2 hits), also measured now.  hbm_frac = memory-side bytes / launch time / 8 TB/s."""
    launches = max(int(st.extend_launches), 1)
    ext_s = st.extend_ms * 1e-3 / launches
    c = cst.extend_counts
    nodes = c.tlas_nodes + c.blas_nodes
    tri_lds, tri_scan = int(cst.extend_tri_wave_lds), int(cst.extend_tri_wave_mem)
    tri_lane = int(c.triangles) - tri_lds - tri_scan
    requested = 32 * nodes + 36 * tri_lane + 36 * tri_scan + 68 * c.instance_visits + 48 * c.rays
    scene_bytes, cache_bytes = pt.scene_bytes()
    gather_l2 = pt.bench_gather(2 << 20, 512, 3)     # lane-steps/s, table inside one XCD's L2
    gather_scene = pt.bench_gather(0, 512, 3)        # table = this scene's nodes + triangles, uniformly random
    gather_mall = pt.bench_gather(192 << 20, 512, 3)  # table beyond L2, inside the Infinity Cache
    achieved = requested / launches / ext_s / 1e9 if ext_s > 0 else 0.0
    peak = gather_l2 * 32 / 1e9
    alg_bytes = c.algorithmic_bytes(112)
    traffic = traffic_src = fabric_frac = None
    try:
        tj = json.load(open(args.traffic_json))
        entries = tj.get("entries", [tj])
        sha = device_sources_sha()
        same = [e for e in entries if e.get("workload") == workload and e.get("paths_in_flight") == n_paths]
        hit = [e for e in same if e.get("device_sources_sha") == sha]
        if hit:
            e = hit[0]
            traffic = e.get("memory_side_bytes_per_extend_launch")
            traffic_src = {k: e.get(k) for k in ("measured_at_head", "fetch_size_bytes", "write_size_bytes",
                                                 "fetch_calibration")}
            traffic_note = "counter passes taken on these very device sources"
            if gather_mall > 0 and ext_s > 0:
                fabric_frac = e["fetch_size_bytes"] / 64.0 / ext_s / gather_mall
        elif same:
            traffic_note = (f"not reported: {os.path.basename(args.traffic_json)} was measured on device sources "
                            f"{same[0].get('device_sources_sha')}, this build is {sha} — run tools/traffic_pass.sh again")
        else:
            traffic_note = f"not reported: {os.path.basename(args.traffic_json)} holds no counter pass for '{workload}'"
    except Exception as e:  # noqa: BLE001 - the line says why the figure is missing
        traffic = None
        traffic_note = f"not reported: {type(e).__name__}: {e}"
    per_ray = {"tlas_nodes": c.tlas_nodes / max(c.rays, 1), "instance_visits": c.instance_visits / max(c.rays, 1),
               "blas_nodes": c.blas_nodes / max(c.rays, 1), "triangles": c.triangles / max(c.rays, 1),
               "triangles_fetched_by_a_lane": tri_lane / max(c.rays, 1),
               "triangles_in_whole_wave_scans_from_lds": tri_lds / max(c.rays, 1),
               "triangles_in_whole_wave_scans_from_memory": tri_scan / max(c.rays, 1)}
    return {
        "bound": "l1_gather", "kernel": "k_extend", "unit": "GB/s",
        "binding_resource": "per-lane dependent gathers: L1 / texture-address path first, L2-miss path second "
                            "(the scene is resident in L2 + Infinity Cache; HBM itself is nearly idle)",
        "achieved": achieved, "peak": peak, "frac": achieved / peak if peak > 0 else None,
        "frac_source": "mirt_bench_gather",
        "peak_source": f"mirt_bench_gather in this process: {gather_l2 / 1e9:.1f} G lane-steps/s x 32 B from a 2 MiB table",
        "fabric_frac": fabric_frac,
        "gather_lane_steps_per_s": {"2MiB_table": gather_l2, "192MiB_table": gather_mall,
                                    f"scene_sized_table_{scene_bytes / 1e6:.0f}MB_uniform": gather_scene},
        "requested_bytes_per_launch": requested / launches,
        "lds_served_triangle_tests_per_launch": tri_lds / launches,
        "lds_served_bytes_equivalent_per_launch": 36 * tri_lds / launches,
        "launch_ms": ext_s * 1e3, "launches": launches,
        "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note, "traffic_key": workload,
        "hbm_peak": HBM_PEAK_GBS,
        "hbm_frac": (traffic / ext_s / 1e9 / HBM_PEAK_GBS) if traffic and ext_s > 0 else None,
        # the traffic model in the reference's struct sizes (64 B per node, 160 B per triangle test, 76 B per
        # instance visit, 112 B per ray): a throughput figure in the reference's units, NOT memory traffic
        "algorithmic_GBs_reference_layout": alg_bytes / launches / ext_s / 1e9 if ext_s > 0 else 0.0,
        "algorithmic_GBs_reference_layout_all_kernels": (alg_bytes + 340.0 * st.shade_events + 236.0 * st.spawned_samples)
        / elapsed / 1e9,
        "node_steps_per_s": nodes / launches / ext_s if ext_s > 0 else 0.0,
        "per_ray": per_ray,
        "kernels": kernel_table(st, cst, classic),
    }


def kernel_table(st, cst, classic=False):
    """Every kernel group of the frame priced the same way: algorithmic bytes in the reference's struct sizes
    (A_extend = 112 B per ray + 64 B per node visit + 76 B per instance visit + 160 B per triangle test; A_connect =
    52 B per shadow ray + the same per visit, with the any-hit walk's own counts; A_shade = 340 B per shade event;
    A_spawn = 236 B per spawned sample) / the group's summed HIP-event time / the 8 TB/s HBM peak.  The two
    traversals are throughput figures in reference units (most of their visits are served from L1 / L2 / LDS); for
    the streaming kernels — extend epilogue, material / uber shade kernels, sample kernel, which read and write
    scattered path records once — it is close to real traffic."""
    rows = []

    def row(name, ms, launches, alg_bytes, note):
        if not launches:
            return
        gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None
        rows.append({"kernels": name, "launches": int(launches), "ms_total": ms, "ms_per_launch": ms / launches,
                     "algorithmic_bytes": alg_bytes, "algorithmic_GBs": gbs,
                     "frac_of_hbm_peak": gbs / HBM_PEAK_GBS if gbs else None, "note": note})

    row("k_extend", st.extend_ms, st.extend_launches, cst.extend_counts.algorithmic_bytes(112),
        "closest-hit walk; reference-layout units, not memory traffic")
    if st.connect_launches:
        row("k_connect", st.connect_ms, st.connect_launches, cst.connect_counts.algorithmic_bytes(52),
            "any-hit walk of the shadow rays (stops at the first hit)")
    stream_ms = st.finish_ms + st.shade_ms + st.sample_ms
    stream_launches = st.finish_launches + st.shade_launches + st.sample_launches
    names = ("k_extend_finish + k_connect_shade (or k_connect_finish + k_material<1..4>) + k_sample" if st.finish_launches else
             "k_finish_shade_classic + k_sample" if classic else "k_finish_shade_uber + k_sample")
    row(names, stream_ms, stream_launches, 340.0 * st.shade_events + 236.0 * st.spawned_samples,
        "streaming kernels: A_shade per shade event + A_spawn per spawned sample")
    return rows


def step_mode(args, pt, info, W, H, src, cam, params, world):
    """The drop-in's own operating point: one mirt_step(1) per frame, as the reference's
    pathtracer_phase_execute submits four dispatches per frame (src/pathtracer_manager.rs:74-124),
    with T = W*H path slots (src/pathtracer.rs:34-38).  A bench step is --frames-per-step frames."""
    import torch
    ppp = args.step_paths_per_pixel
    n_paths = W * H * ppp  # Pathtracer.threads = dims.0 * dims.1 in the reference (ppp = 1)
    if args.rng == "per-slot":
        from raytracer_amd import ffi
        params.rng_mode = ffi.RNG_PER_SLOT
    pt.create_state(W, H, n_paths, src, params)
    pt.set_camera(cam)
    if args.rng == "per-slot":
        # PathtracerState::new: every word of every slot's state uniform in [1000, 2^32)
        pt.set_rng(np.random.default_rng(info.seed).integers(1000, 2 ** 32, size=(n_paths, 4), dtype=np.uint64).astype("<u4"))
    F = args.frames_per_step
    for _ in range(args.warmup):
        for _ in range(F):
            pt.step(1)
    pt.synchronize()
    pt.stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(F):
            pt.step(1)
    pt.synchronize()
    elapsed = time.perf_counter() - t0
    st = pt.stats()
    rays = int(st.extend_rays + st.connect_rays)
    frames = int(st.frames)
    # the kernels' own time per frame, from HIP events on the library's stream, in a second untimed pass
    pt.stats_reset()
    pt.set_event_timing(True)
    for _ in range(F):
        pt.step(1)
    pt.synchronize()
    pt.set_event_timing(False)
    ev = pt.stats()
    # a host that waits for every frame (reads the image back each time) pays the launch latency too
    t1 = time.perf_counter()
    for _ in range(F):
        pt.step(1)
        pt.synchronize()
    sync_ms_per_frame = (time.perf_counter() - t1) / F * 1e3
    kernel_ms_per_frame = (ev.extend_ms + ev.finish_ms + ev.shade_ms + ev.sample_ms + ev.connect_ms) / max(int(ev.frames), 1)
    wall_ms_per_frame = elapsed / max(frames, 1) * 1e3
    geometry = GEOMETRY[info.procedural_dragon]
    out = {
        "metric": "Mrays/s (extend+shadow), per-frame stepping", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config{args.config}: {SCENES[args.config]} {W}x{H}, {info.max_bounces} bounces, "
                               f"mirt_step(1) per frame, {F} frames per step, {args.rng} random streams, "
                               f"{'per-material' if info.shade_mode else 'uber'} shading",
                   "geometry": geometry, "paths_in_flight": n_paths, "parallelism": "one GPU"},
        "frames_per_s": frames / elapsed, "wall_ms_per_frame": wall_ms_per_frame,
        "kernel_ms_per_frame": kernel_ms_per_frame, "wall_ms_per_frame_synchronised_every_frame": sync_ms_per_frame,
        "host_and_launch_overhead_us_per_frame": (wall_ms_per_frame - kernel_ms_per_frame) * 1e3,
        "kernel_ms_per_frame_by_kernel": {"extend": ev.extend_ms / max(int(ev.frames), 1), "shade": ev.shade_ms / max(int(ev.frames), 1),
                                          "sample": ev.sample_ms / max(int(ev.frames), 1),
                                          "extend_finish": ev.finish_ms / max(int(ev.frames), 1),
                                          "connect": ev.connect_ms / max(int(ev.frames), 1)},
        "spp_per_s": int(st.finished_samples) / (W * H) / elapsed, "rays": rays,
    }
    print(json.dumps(out), flush=True)


def build_workload(args, geometry):
    """Scene buffers, sample sources, camera and parameters of the configuration being measured."""
    from raytracer_amd import host, pathtracer
    arrays, info = host.build_config_scene(args.config, standin=geometry)
    W = args.width or info.width
    H = args.height or info.height
    if args.mode == "step":
        W, H = args.width or 512, args.height or 512
    if args.spp_per_step <= 0:
        args.spp_per_step = int(info.spp) if 64 <= int(info.spp) <= 256 else 64
    src = host.make_sample_sources(W, H, info.seed, shuffle=2)
    cam = host.default_camera(W, H)
    params = pathtracer.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=info.shade_mode,
                                       nee=info.nee, stage_kernels=1 if args.stage_kernels else 0)
    return arrays, info, W, H, src, cam, params


def timed_render(pt, steps, warmup, spp_step, first_sample, before=None, after=None):
    """`warmup` untimed and `steps` timed mirt_render_spp(spp_step) passes on one context; `before` runs between
    the two (barrier), `after` inside the timed region behind the last pass (the reduction).  Returns
    (seconds, stats, whatever `after` returned)."""
    pt.set_sample_range(first_sample)
    for _ in range(warmup):
        pt.render_spp(spp_step)
    pt.synchronize()
    pt.stats_reset()
    pt.set_event_timing(True)
    if before:
        before()
    pt.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pt.render_spp(spp_step)
    pt.synchronize()
    extra = after() if after else None
    elapsed = time.perf_counter() - t0
    pt.set_event_timing(False)
    return elapsed, pt.stats(), extra


REPLAY_STEPS = 4


def replay_counts(pt, ranges, st):
    """Visit counts of the timed samples: the same sample numbers rendered again, untimed, with the counting
    build of the traversal kernel.  `st` holds the timed run's statistics: a replay of every timed step must trace
    exactly its rays.  With more than REPLAY_STEPS timed steps only the first REPLAY_STEPS are replayed (every step
    renders the same number of samples of the same scene, so per-ray figures do not move) and the tallies are
    scaled to the timed ray counts; the roofline block then says so (`counts_replayed_steps`)."""
    pt.stats_reset()
    pt.set_counting(True)
    for first, n in ranges[:REPLAY_STEPS]:
        pt.set_sample_range(first)
        pt.render_spp(n)
    pt.set_counting(False)
    cst = pt.stats()
    if len(ranges) <= REPLAY_STEPS:
        assert cst.extend_counts.rays == st.extend_rays, (cst.extend_counts.rays, st.extend_rays)
    else:
        for counts, timed_rays, waves in ((cst.extend_counts, int(st.extend_rays), ("extend_tri_wave_lds", "extend_tri_wave_mem")),
                                          (cst.connect_counts, int(st.connect_rays), ())):
            if not counts.rays:
                continue
            k = timed_rays / counts.rays
            assert abs(k * min(len(ranges), REPLAY_STEPS) / len(ranges) - 1.0) < 0.02, (k, len(ranges))
            for name in ("rays", "tlas_nodes", "instance_visits", "blas_nodes", "triangles"):
                setattr(counts, name, int(round(getattr(counts, name) * k)))
            for name in waves:
                setattr(cst, name, int(round(getattr(cst, name) * k)))
        cst.shade_events = st.shade_events
        cst.spawned_samples = st.spawned_samples
    cst_steps = min(len(ranges), REPLAY_STEPS)
    return cst, cst_steps


def workload_name(args, W, H, geometry):
    # (the key of a counter pass in profiles/traffic.json: traffic per launch depends on all of these)
    return f"config{args.config} {W}x{H} {args.spp_per_step} spp per step" + ("" if geometry == "knot" else f" {geometry}")


def secondary_line(args, local_rank):
    """The headline configuration on the second stand-in geometry (no pathological 2 888-triangle leaf), two
    steps after one warm-up, with its own roofline block: a second driver-visible figure beside the knot's."""
    from raytracer_amd import pathtracer
    geometry = "suzanne"
    arrays, info, W, H, src, cam, params = build_workload(args, geometry)
    n_paths = W * H * args.paths_per_pixel
    with pathtracer.Pathtracer(local_rank) as pt:
        pt.upload_scene(arrays)
        pt.create_state(W, H, n_paths, src, params)
        pt.set_camera(cam)
        steps, warmup, spp_step = 2, 1, args.spp_per_step
        elapsed, st, _ = timed_render(pt, steps, warmup, spp_step, 0)
        cst, cst_steps = replay_counts(pt, [((warmup + k) * spp_step, spp_step) for k in range(steps)], st)
        rays = int(st.extend_rays + st.connect_rays)
        roof = extend_roofline(pt, st, cst, elapsed, args, workload_name(args, W, H, geometry), n_paths)
        roof["counts_replayed_steps"] = cst_steps
    keep = ("bound", "kernel", "unit", "achieved", "peak", "frac", "frac_source", "fabric_frac", "hbm_frac", "traffic",
            "traffic_note", "launch_ms", "launches", "per_ray", "counts_replayed_steps")
    return {"metric": "Mrays/s (extend+shadow), dragon 1080p", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "config": {"workload": f"config{args.config}: {SCENES[args.config]} {W}x{H}, {info.max_bounces} bounces, "
                                   f"{spp_step} spp per step, uber shading",
                       "geometry": GEOMETRY[info.procedural_dragon], "paths_in_flight": n_paths},
            "roofline": {k: roof.get(k) for k in keep}}


def group_mode(args):
    """One process, N GPUs: the same workload through mirt_group_* (PathtracerGroup) — scene, state and camera
    replicated, every render call's samples split over the members on one host thread each, accumulators added
    into member 0 by peer copies inside the timed region.  Weak scaling: spp_per_step samples per member and step."""
    from raytracer_amd import pathtracer
    n = args.gpus
    arrays, info, W, H, src, cam, params = build_workload(args, args.geometry)
    n_paths = W * H * args.paths_per_pixel
    spp_step = args.spp_per_step
    with pathtracer.PathtracerGroup([device_of(i) for i in range(n)]) as grp:
        grp.upload_scene(arrays)
        grp.create_state(W, H, n_paths, src, params)
        grp.set_camera(cam)
        for _ in range(args.warmup):
            grp.render_spp(spp_step * n)
        grp.reduce()
        for m in grp.members:
            m.synchronize()
            m.stats_reset()
            m.set_event_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            grp.render_spp(spp_step * n)
        t1 = time.perf_counter()
        grp.reduce()
        elapsed = time.perf_counter() - t0
        reduce_s = time.perf_counter() - t1
        stats = []
        for m in grp.members:
            m.set_event_timing(False)
            stats.append(m.stats())
        rays = sum(int(s.extend_rays + s.connect_rays) for s in stats)
        samples = sum(int(s.finished_samples) for s in stats)
        per_member = [int(s.extend_rays + s.connect_rays) / elapsed / 1e6 for s in stats]
        # member 0 took the first spp_step sample numbers of every call
        m0 = grp.members[0]
        cst, cst_steps = replay_counts(m0, [((args.warmup + k) * spp_step * n, spp_step) for k in range(args.steps)], stats[0])
        roofline = extend_roofline(m0, stats[0], cst, elapsed, args, workload_name(args, W, H, args.geometry), n_paths,
                                   classic=bool(info.shade_mode))
        roofline["counts_replayed_steps"] = cst_steps
        st = stats[0]
    out = result_line(args, info, W, H, n_paths, n, elapsed, rays, samples, reduce_s, st, roofline, None,
                      per_member, f"spp split x{n}, one process (mirt_group), peer-copy reduce to member 0")
    print(json.dumps(out), flush=True)


def result_line(args, info, W, H, n_paths, world, elapsed, rays, samples, reduce_s, st, roofline, cpu, per_rank,
                parallelism):
    geometry = GEOMETRY[info.procedural_dragon] if args.config >= 3 else "the reference's own assets (suzanne / teapot)"
    return {
        "metric": "Mrays/s (extend+shadow), dragon 1080p",
        "value": rays / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config{args.config}: {SCENES[args.config]} {W}x{H}, "
                               f"{info.max_bounces} bounces, {args.spp_per_step} spp per step per GPU, "
                               f"{'per-material' if info.shade_mode else 'uber'} shading",
                   "geometry": geometry, "paths_in_flight": n_paths, "parallelism": parallelism},
        "spp_per_s": samples / (W * H) / elapsed,
        "rays": rays,
        "reduce_ms": reduce_s * 1e3,
        "per_rank_mrays_per_s": {"min": min(per_rank), "max": max(per_rank)},
        "kernel_ms": {"extend": st.extend_ms, "extend_finish": st.finish_ms, "shade": st.shade_ms,
                      "sample": st.sample_ms, "connect": st.connect_ms},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.group:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "pid": os.getpid(), "gpus": args.gpus,
                          **{k: os.environ.get(k) for k in RANK_ENV}}), flush=True)
        if os.environ.get("MIRT_BENCH_DRY_FAIL_RANK") == str(rank):  # lets a test see the launcher report a dead rank
            sys.exit(3)
        return
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.group and args.gpus > 1 and world == 1:
        group_mode(args)
        return
    dist = None
    import torch
    # MIRT_BENCH_DIST=1 takes the collective path with a single rank too (a rehearsal
    # of the multi-GPU code on a one-GPU box)
    if world > 1 or os.environ.get("MIRT_BENCH_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(device_of(local_rank))
        backend = os.environ.get("MIRT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_of(local_rank)))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = torch.device("cuda", device_of(local_rank))

    from raytracer_amd import distributed, pathtracer

    arrays, info, W, H, src, cam, params = build_workload(args, args.geometry)
    pt = pathtracer.Pathtracer(device_of(local_rank))
    pt.upload_scene(arrays)
    if args.mode == "step":
        step_mode(args, pt, info, W, H, src, cam, params, world)
        pt.close()
        return
    n_paths = W * H * args.paths_per_pixel
    pt.create_state(W, H, n_paths, src, params)
    pt.set_camera(cam)

    spp_step = args.spp_per_step
    total_steps = args.warmup + args.steps
    # rank r owns sample numbers [r * total, (r + 1) * total) * spp_step
    first = rank * total_steps * spp_step

    def before():
        if dist is not None:
            # warm the collective too (its result is discarded: the accumulators are reloaded with the sum of
            # all ranks' warm-up samples, which the timed region then adds to)
            distributed.reduce_device_accumulators(pt, device)
            dist.barrier()
        torch.cuda.synchronize(device)

    def after():
        reduce_s = 0.0
        if dist is not None:
            reduce_s = distributed.reduce_device_accumulators(pt, device)
            dist.barrier()
        torch.cuda.synchronize(device)
        return reduce_s

    elapsed, st, reduce_s = timed_render(pt, args.steps, args.warmup, spp_step, first, before, after)
    rays = int(st.extend_rays + st.connect_rays)
    samples = int(st.finished_samples)
    mine = rays / elapsed / 1e6
    per_rank = [mine, mine]

    if dist is not None:
        t = torch.tensor([elapsed, float(rays), float(samples), reduce_s, mine, -mine], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        reduce_s = float(tmax[3])
        rays = int(tsum[1])
        samples = int(tsum[2])
        per_rank = [-float(tmax[5]), float(tmax[4])]

    if rank == 0:
        # ---- roofline of the dominant kernel: replay the timed samples with visit counting
        cst, cst_steps = replay_counts(pt, [(first + (args.warmup + k) * spp_step, spp_step) for k in range(args.steps)], st)
        roofline = extend_roofline(pt, st, cst, elapsed, args, workload_name(args, W, H, args.geometry), n_paths,
                                   classic=bool(info.shade_mode))
        roofline["counts_replayed_steps"] = cst_steps
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the host-core baseline belongs to the one-GPU line only
            cpu = cpu_baseline(arrays, info, W, H, src, cam, args.cpu_seconds)
        out = result_line(args, info, W, H, n_paths, world, elapsed, rays, samples, reduce_s, st, roofline, cpu,
                          per_rank, f"spp split x{world}" + (", one process per GPU, RCCL all-reduce" if world > 1 else ""))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pt.close()
    if rank == 0:
        # the second stand-in geometry, after this context has released its gigabytes of path state
        if (world == 1 and args.config == 3 and args.geometry == "knot" and args.mode == "render"
                and not args.no_secondary and args.width == 0 and args.height == 0):
            out["secondary"] = secondary_line(args, device_of(local_rank))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
