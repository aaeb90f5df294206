"""Regenerates the committed fixtures under tests/golden/.

  suzanne.mirtmesh, teapot.mirtmesh   parsed first model of the reference's OBJ assets
                                      (needs /root/reference; skipped when it is absent)
  kat.json                            known-answer values derived from the TEXT of the
                                      reference's formulas (RNG) and builders (rect, cube)
  golden.npz                          ray sets with their hit ids, and accumulated sums of
                                      small renders, produced by the CPU oracle

The reference itself has no tests or golden vectors and cannot be run here, so
these fixtures pin this repository's restatement against regressions; they are
not outputs of the reference.  Run from the repository root: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.join(ROOT, "tests", "golden")

from raytracer_amd import ffi, host  # noqa: E402
from oracle import host_ref, pyoracle as po  # noqa: E402


def rng_kat(seed, n):
    """random.slang:9-30 evaluated with Python integers (no shared code with the oracle)."""
    x, y, z, w = seed
    M = 0xFFFFFFFF

    def taus(v, s1, s2, s3, m):
        b = ((((v << s1) & M) ^ v) >> s2)
        return ((((v & m) << s3) & M) ^ b)

    out = []
    for _ in range(n):
        x = taus(x, 13, 19, 12, 4294967294)
        y = taus(y, 2, 25, 4, 4294967288)
        z = taus(z, 3, 11, 17, 4294967280)
        w = (1664525 * w + 1013904223) & M
        val = np.float32(2.3283064365387e-10) * np.float32(x ^ y ^ z ^ w)
        out.append({"state": [x, y, z, w], "value_bits": int(np.float32(val).view(np.uint32))})
    return out


def main():
    ref_assets = "/root/reference/assets"
    if os.path.isdir(ref_assets):
        L = ffi.host_lib()
        for name in ("suzanne", "teapot"):
            rc = L.mirth_obj_to_model_file(f"{ref_assets}/{name}.obj".encode(), f"{HERE}/{name}.mirtmesh".encode())
            assert rc == 0, name

    kat = {"rng": {}, "bvh": {}}
    for seed in ([1000, 1000, 1000, 1000], [123456789, 362436069, 521288629, 88675123]):
        kat["rng"][",".join(map(str, seed))] = rng_kat(seed, 8)

    # rect and cube through the pure-Python builder (oracle/host_ref.py)
    s = host.Scene()
    for name, mid in (("rect", s.load_rect()), ("cube", s.load_cube())):
        # literal geometry of src/mesh.rs:415-536, read back from the host library
        # BEFORE its reorder is irrelevant: the faces are re-sorted into file order
        pos, nrm, faces, _ = s.mesh(mid)
        order = np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))
        faces0 = faces[order]
        nodes, faces_py = host_ref.build_blas([np.asarray(p, np.float32) for p in pos], faces0.tolist())
        kat["bvh"][name] = {
            "faces": [list(map(int, f)) for f in faces_py],
            "nodes": [[list(lb), list(ub), left, skip, leaf, start, end]
                      for lb, ub, left, skip, leaf, start, end in host_ref.nodes_as_records(nodes)],
        }
    s.close()
    kat["assets"] = {"suzanne": {"positions": 507, "triangles": 968}, "teapot": {"positions": 3644, "triangles": 6320}}
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # oracle-produced vectors
    out = {}
    for cfg, (W, H, spp) in ((1, (64, 64, 2)), (2, (64, 36, 1))):
        arrays, info = host.build_config_scene(cfg, asset_dir=HERE)
        sc = po.Scene(arrays)
        src = host.make_sample_sources(W, H, info.seed)
        cam = host.default_camera(W, H)
        for mode in (0, 1):
            prm = po.default_params(max_bounces=info.max_bounces, seed=info.seed, shade_mode=mode, nee=mode)
            s4, cnt, st = sc.render(cam, src, W, H, prm, 0, spp)
            out[f"cfg{cfg}_mode{mode}_sum4"] = s4
            out[f"cfg{cfg}_mode{mode}_count"] = cnt
            out[f"cfg{cfg}_mode{mode}_rays"] = np.array([st.extend_rays, st.connect_rays, st.shade_events], "<u8")
        prm = po.default_params(max_bounces=info.max_bounces, seed=info.seed)
        for bounce in (0, 1):
            rays = sc.bounce_rays(cam, src, W, H, prm, 0, bounce)
            rays = rays[rays["last_inst"] != ffi.NO_HIT]
            out[f"cfg{cfg}_bounce{bounce}_rays"] = rays.view("<u4").reshape(-1, 8)
            out[f"cfg{cfg}_bounce{bounce}_hits"] = sc.trace_closest(rays).view("<u4").reshape(-1, 4)
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
