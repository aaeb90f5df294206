"""Mutation fuzzing of the host library's parsers of untrusted input — glTF / GLB documents (with their
base64 and external buffers and PNG images), PNG images, OBJ text, model files — under AddressSanitizer and
UndefinedBehaviorSanitizer.  CPU only (host code): builds libmirt_host and tools/fuzz_host_driver.c with
-fsanitize=address,undefined into build/fuzz/, writes valid documents with the writers of tests/test_gltf.py,
mutates them (byte flips, truncation, splices, numbers in the JSON replaced by hostile ones, indices bent) and
runs the driver on each with a time limit.  A sanitizer report, a crash or a timeout is a finding and is kept
under build/fuzz/findings/.

    python tools/fuzz_host.py [--runs 3000] [--seed 1] [--jobs 8]
"""
import argparse
import base64
import concurrent.futures
import glob
import json
import os
import pathlib
import random
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "fuzz")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
HOSTILE = ["-1", "0", "1", "2", "255", "65535", "65536", "4294967295", "4294967296", "2147483647", "-2147483648",
           "1e999", "-1e999", "1e-999", "9007199254740993", "0.5", "-0.0", "1e308", "18446744073709551616", "NaN"]


def build():
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libmirt_host_asan.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "raytracer_amd", "csrc", "host", "*.cpp")))
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", *SAN,
                           "-I", os.path.join(ROOT, "include"), *srcs, "-o", lib])
    drv = os.path.join(OUT, "fuzz_host_driver")
    subprocess.check_call(["gcc", "-std=c11", *SAN, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "fuzz_host_driver.c"), lib, "-Wl,-rpath," + OUT, "-o", drv])
    return drv


def corpus():
    """(kind, name, bytes, sidecar files) of valid inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import test_gltf as tg
    seeds = []
    base = pathlib.Path(OUT) / "corpus"
    shutil.rmtree(base, ignore_errors=True)
    for mode in ("glb", "embedded", "external"):
        d = base / mode
        d.mkdir(parents=True)
        path, _, _ = tg.build_document(d, mode)
        side = {os.path.basename(p): open(p, "rb").read() for p in glob.glob(str(d / "*")) if p != path}
        seeds.append(("gltf", os.path.basename(path), open(path, "rb").read(), side))
    img = (np.arange(7 * 5 * 4).reshape(5, 7, 4) * 3 % 256).astype("u1")
    for depth, ctype, lvl in ((8, 6, 6), (8, 2, 0), (16, 6, 9), (8, 0, 6), (8, 3, 6)):
        try:
            if ctype == 6:
                rows = [img[y].tobytes() if depth == 8 else img[y].astype(">u2").tobytes() for y in range(5)]
                png = tg.write_png(rows, 7, 5, depth, ctype, level=lvl)
            elif ctype == 2:
                png = tg.write_png([img[y, :, :3].tobytes() for y in range(5)], 7, 5, depth, ctype, level=lvl)
            elif ctype == 0:
                png = tg.write_png([img[y, :, 0].tobytes() for y in range(5)], 7, 5, depth, ctype, level=lvl)
            else:
                pal = bytes(range(48))
                png = tg.write_png([bytes(int(v) % 16 for v in img[y, :, 0]) for y in range(5)], 7, 5, depth, ctype,
                                   level=lvl, palette=pal, trns=bytes([0, 128, 255]))
            seeds.append(("png", f"img_{depth}_{ctype}.png", png, {}))
        except Exception as e:  # a writer variant the test helpers do not offer
            print("corpus: skipped png", depth, ctype, e)
    obj = "\n".join(["# a quad, a fan and negative indices, normals on every corner", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0",
                     "v 0.5 0.5 1", "vn 0 0 1", "vn 0 1 0", "vn 1 0 0", "vn 0 0 -1", "vn 0 -1 0", "vt 0 0", "vt 1 1",
                     "f 1//1 2//2 3//3 4//4", "f 1//1 2//1 5//2", "f -1/1/1 -2/2/2 -3/1/1", "f 1/1/1 2/2/2 3/1/3 4/2/4 5/1/5", ""]).encode()
    plain = "\n".join(["o no normals", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 0.5 1e0", "f 1 2 3 4", "f 1 2 5",
                       "f -1 -2 -3", "f 1/1 2/2 3/1 4/2 5/1", "vt 0 0", "vt 1 1", ""]).encode()
    seeds.append(("obj", "plain.obj", plain, {}))
    seeds.append(("obj", "mesh.obj", obj, {}))
    for m in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.mirtmesh")))[:1]:
        seeds.append(("model", os.path.basename(m), open(m, "rb").read()[:200000], {}))
    return seeds


def mutate_json_text(rng, text):
    nums = list(re.finditer(r"(?<![\w\"])-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?", text))
    for _ in range(rng.randint(1, 4)):
        kind = rng.random()
        if nums and kind < 0.6:
            m = rng.choice(nums)
            text = text[:m.start()] + rng.choice(HOSTILE) + text[m.end():]
            nums = list(re.finditer(r"(?<![\w\"])-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?", text))
        elif kind < 0.75:
            i = rng.randrange(len(text))
            text = text[:i] + rng.choice(['{', '}', '[', ']', '"', ',', ':', 'null', 'true', '[[[[[[[[', '\\u0000', '\\']) + text[i:]
        elif kind < 0.9:
            i, j = sorted(rng.randrange(len(text)) for _ in range(2))
            text = text[:i] + text[j:]
        else:
            i, j = sorted(rng.randrange(len(text)) for _ in range(2))
            text = text[:j] + text[i:j] + text[j:]
    return text


def mutate_bytes(rng, data):
    b = bytearray(data)
    for _ in range(rng.randint(1, 6)):
        kind = rng.random()
        if not b:
            break
        if kind < 0.5:
            i = rng.randrange(len(b))
            b[i] = rng.choice([0, 1, 0x7F, 0x80, 0xFF, b[i] ^ (1 << rng.randrange(8)), rng.randrange(256)])
        elif kind < 0.65:
            i = rng.randrange(max(len(b) - 4, 1))
            b[i:i + 4] = rng.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80", b"\x01\x00\x00\x00"])
        elif kind < 0.8:
            del b[rng.randrange(len(b)):]
        elif kind < 0.9:
            i, j = sorted(rng.randrange(len(b)) for _ in range(2))
            del b[i:j]
        else:
            i, j = sorted(rng.randrange(len(b)) for _ in range(2))
            b[j:j] = b[i:j][:4096]
    return bytes(b)


def mutate(rng, kind, name, data):
    if kind == "gltf" and name.endswith(".gltf") and rng.random() < 0.8:
        text = data.decode("utf-8", "replace")
        if rng.random() < 0.3:  # bend the base64 payloads as bytes
            def bend(m):
                try:
                    raw = base64.b64decode(m.group(2))
                except Exception:
                    return m.group(0)
                return m.group(1) + base64.b64encode(mutate_bytes(rng, raw)).decode()
            text = re.sub(r"(base64,)([A-Za-z0-9+/=]+)", lambda m: bend(m) if rng.random() < 0.5 else m.group(0), text)
        return mutate_json_text(rng, text).encode("utf-8", "replace")
    if kind == "gltf" and name.endswith(".glb") and len(data) > 20 and rng.random() < 0.6:
        # mutate the JSON chunk as text and re-frame it, so that the fuzzing reaches past the container checks
        import struct
        jlen = struct.unpack_from("<I", data, 12)[0]
        js = mutate_json_text(rng, data[20:20 + jlen].decode("utf-8", "replace")).encode("utf-8", "replace")
        js += b" " * (-len(js) % 4)
        rest = data[20 + jlen:]
        if rng.random() < 0.3:
            rest = mutate_bytes(rng, rest)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + rest
        return b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body
    if kind == "obj" and rng.random() < 0.7:
        return mutate_json_text(rng, data.decode("utf-8", "replace")).encode("utf-8", "replace")
    return mutate_bytes(rng, data)


def run_one(job):
    drv, kind, path = job
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=4096",
               UBSAN_OPTIONS="print_stacktrace=1")
    try:
        r = subprocess.run([drv, kind, path], capture_output=True, timeout=20, env=env)
        bad = r.returncode != 0
        return path, ("exit %d" % r.returncode) if bad else None, r.stderr[-3000:].decode("utf-8", "replace") if bad else ""
    except subprocess.TimeoutExpired:
        return path, "timeout", ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--jobs", type=int, default=8)
    args = ap.parse_args()
    drv = build()
    seeds = corpus()
    rng = random.Random(args.seed)
    work = pathlib.Path(OUT) / "work"
    shutil.rmtree(work, ignore_errors=True)
    findings = pathlib.Path(OUT) / "findings"
    findings.mkdir(exist_ok=True)
    jobs = []
    # the unmutated corpus first: it must load
    for k, (kind, name, data, side) in enumerate(seeds):
        d = work / f"seed{k}"
        d.mkdir(parents=True)
        for sn, sd in side.items():
            (d / sn).write_bytes(sd)
        (d / name).write_bytes(data)
        jobs.append((drv, kind, str(d / name)))
        r = subprocess.run([drv, kind, str(d / name)], capture_output=True, timeout=60)
        if r.returncode != 0 or b"refused" in r.stderr:
            print(f"corpus: {kind} {name} does not load as it is:", r.stderr.decode()[-300:])
    for i in range(args.runs):
        kind, name, data, side = rng.choice(seeds)
        d = work / f"m{i}"
        d.mkdir(parents=True)
        for sn, sd in side.items():
            (d / sn).write_bytes(mutate_bytes(rng, sd) if rng.random() < 0.3 else sd)
        (d / name).write_bytes(mutate(rng, kind, name, data))
        jobs.append((drv, kind, str(d / name)))
    n_bad = 0
    with concurrent.futures.ThreadPoolExecutor(args.jobs) as ex:
        for path, verdict, log in ex.map(run_one, jobs):
            if verdict:
                n_bad += 1
                dst = findings / (os.path.basename(os.path.dirname(path)) + "_" + os.path.basename(path))
                shutil.copy(path, dst)
                (findings / (dst.name + ".log")).write_text(verdict + "\n" + log)
                print("FINDING", verdict, path)
                print(log[-1200:])
    print(f"{len(jobs)} inputs ({len(seeds)} valid, {args.runs} mutated): {n_bad} findings")
    shutil.rmtree(work, ignore_errors=True)
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
