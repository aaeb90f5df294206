"""How much does ray order matter to the closest-hit engine?  Traces the same
bounce-1 ray set of configuration 3 in queue order, shuffled and sorted by a
Morton key of the origin plus the direction octant; primary rays for reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import host, pathtracer

arrays, info = host.build_config_scene(3)
W, H = 1920, 1080
src = host.make_sample_sources(W, H, info.seed, shuffle=2)
cam = host.default_camera(W, H)
pt = pathtracer.Pathtracer(0)
pt.upload_scene(arrays)
PPP = int(os.environ.get("PPP", "2"))
pt.create_state(W, H, W * H * PPP, src, pathtracer.default_params(max_bounces=8, seed=info.seed))
pt.set_camera(cam)


def rate(name, rays):
    ms = pt.bench_trace_closest(rays, repeats=5)
    print(f"{name:<34} {len(rays):>9} rays  {len(rays) / ms / 1e3:8.1f} Mrays/s", flush=True)


def morton_key(rays):
    p = np.stack([rays["pos"][:, 0], rays["pos"][:, 1], rays["pos"][:, 2]], 1) if rays["pos"].ndim == 2 else None
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint64), 0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    m = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    d = rays["dir"]
    octant = (d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2)
    return (octant << 30) | m


pt.step(1)
r1 = pt.dump_extension_rays(W * H * PPP)
print(r1.dtype)
rate("bounce 1, queue order", r1)
rng = np.random.default_rng(1)
rate("bounce 1, shuffled", r1[rng.permutation(len(r1))])
rate("bounce 1, sorted octant+morton", r1[np.argsort(morton_key(r1), kind="stable")])
pt.step(1)
r2 = pt.dump_extension_rays(W * H * PPP)
rate("frame 2 queue (bounce 2 + new primary)", r2)
rate("frame 2, sorted", r2[np.argsort(morton_key(r2), kind="stable")])


def chunk_shuffled(rays, chunk):
    """Sorted rays, then whole chunks of `chunk` consecutive rays in random order: coherent
    inside a wave's share of work, no expensive region left for the end of the launch."""
    s = rays[np.argsort(morton_key(rays), kind="stable")]
    n = len(s) // chunk
    order = np.random.default_rng(2).permutation(n)
    idx = (order[:, None] * chunk + np.arange(chunk)[None, :]).reshape(-1)
    return np.concatenate([s[idx], s[n * chunk:]])


for chunk in (128, 1024, 8192, 65536):
    rate(f"bounce 1, sorted, chunks of {chunk} shuffled", chunk_shuffled(r1, chunk))
rate("frame 2, sorted, chunks of 128 shuffled", chunk_shuffled(r2, 128))
