"""The glTF 2.0 loader of the host scene model (csrc/host/gltf.cpp) and its PNG decoder.

The reference leaves its glTF loader commented out (src/scenes.rs:332-399) and ships no glTF
file, so there is nothing of the reference's to compare with: every file here is written by the
test (JSON + buffers with struct / base64, PNGs with zlib), and what the loader hands to bind()
is compared with what the test put in."""
import base64
import json
import os
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import ffi, host


# ---------------------------------------------------------------- PNG writing

def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(rows, width, height, depth, ctype, level=6, filters=(0, 1, 2, 3, 4), palette=None, trns=None, split=None):
    """rows: `height` byte strings of packed samples (PNG layout).  Every row gets one of `filters` in turn."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = filters[y % len(filters)]
        out = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            out[i] = (x - pred) & 255
        raw.append(f)
        raw += out
        prev = row
    comp = zlib.compress(bytes(raw), level)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    if palette is not None:
        png += _chunk(b"PLTE", bytes(palette))
    if trns is not None:
        png += _chunk(b"tRNS", bytes(trns))
    if split:  # several IDAT chunks
        for i in range(0, len(comp), split):
            png += _chunk(b"IDAT", comp[i:i + split])
    else:
        png += _chunk(b"IDAT", comp)
    return png + _chunk(b"IEND", b"")


def texels_of(scene):
    arrays = scene.bind()
    return arrays["textures"], arrays["texels"]


def _scene_with_rect():
    s = host.Scene()
    s.spawn(host.make_transform(), s.add_material(host.make_material()), s.load_rect())
    return s


def rgba(r, g, b, a=255):
    return (np.asarray(r, np.uint32) | (np.asarray(g, np.uint32) << 8) | (np.asarray(b, np.uint32) << 16) |
            (np.asarray(a, np.uint32) << 24))


@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_rgb_and_rgba_8_and_16_bit_every_filter(level):
    rng = np.random.default_rng(level)
    W, H = 37, 11  # an odd width; eleven rows run through the five filters twice
    s = _scene_with_rect()
    want = []
    for ctype, ch in ((2, 3), (6, 4)):
        for depth in (8, 16):
            img = rng.integers(0, 1 << depth, (H, W, ch), dtype=np.uint32)
            if level == 9:
                img[:, :, :] = img[:, :1, :]  # long matches for the copy path of inflate
            rows = [img[y].astype(">u2" if depth == 16 else "u1").tobytes() for y in range(H)]
            s.add_texture_png(write_png(rows, W, H, depth, ctype, level=level, split=97 if depth == 16 else None))
            v8 = img >> (depth - 8)
            want.append(rgba(v8[..., 0], v8[..., 1], v8[..., 2], v8[..., 3] if ch == 4 else 255).reshape(-1))
    tex, texels = texels_of(s)
    assert len(tex) == 4 and all(t["width"] == W and t["height"] == H for t in tex)
    assert np.array_equal(texels, np.concatenate(want))
    s.close()


def test_png_grey_palette_and_transparency():
    rng = np.random.default_rng(5)
    s = _scene_with_rect()
    want = []
    W, H = 13, 5
    # grey at 1, 2, 4, 8 and 16 bits
    for depth in (1, 2, 4, 8, 16):
        img = rng.integers(0, 1 << depth, (H, W), dtype=np.uint32)
        if depth < 8:
            rows = []
            for y in range(H):
                bits = "".join(format(int(v), f"0{depth}b") for v in img[y])
                bits += "0" * (-len(bits) % 8)
                rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
            v8 = img * 255 // ((1 << depth) - 1)
        else:
            rows = [img[y].astype(">u2" if depth == 16 else "u1").tobytes() for y in range(H)]
            v8 = img >> (depth - 8)
        s.add_texture_png(write_png(rows, W, H, depth, 0, filters=(0,) if depth < 8 else (4, 3)))
        want.append(rgba(v8, v8, v8).reshape(-1))
    # grey + alpha
    img = rng.integers(0, 256, (H, W, 2), dtype=np.uint32)
    s.add_texture_png(write_png([img[y].astype("u1").tobytes() for y in range(H)], W, H, 8, 4))
    want.append(rgba(img[..., 0], img[..., 0], img[..., 0], img[..., 1]).reshape(-1))
    # palette of five entries, three of them with their own alpha; 8 and 4 bits per index
    pal = rng.integers(0, 256, (5, 3), dtype=np.uint32)
    alpha = np.array([0, 128, 255, 255, 255], np.uint32)
    idx = rng.integers(0, 5, (H, W), dtype=np.uint32)
    s.add_texture_png(write_png([idx[y].astype("u1").tobytes() for y in range(H)], W, H, 8, 3, palette=pal.astype("u1").reshape(-1),
                                trns=alpha[:3].astype("u1")))
    want.append(rgba(pal[idx, 0], pal[idx, 1], pal[idx, 2], alpha[idx]).reshape(-1))
    rows = []
    for y in range(H):
        bits = "".join(format(int(v), "04b") for v in idx[y])
        bits += "0" * (-len(bits) % 8)
        rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    s.add_texture_png(write_png(rows, W, H, 4, 3, filters=(0,), palette=pal.astype("u1").reshape(-1)))
    want.append(rgba(pal[idx, 0], pal[idx, 1], pal[idx, 2]).reshape(-1))
    # a colour key: RGB 8-bit with one colour declared transparent
    img = rng.integers(0, 4, (H, W, 3), dtype=np.uint32)
    key = img[2, 3]
    s.add_texture_png(write_png([img[y].astype("u1").tobytes() for y in range(H)], W, H, 8, 2,
                                trns=struct.pack(">HHH", *[int(k) for k in key])))
    a = np.where((img == key).all(-1), 0, 255)
    want.append(rgba(img[..., 0], img[..., 1], img[..., 2], a).reshape(-1))
    _, texels = texels_of(s)
    assert np.array_equal(texels, np.concatenate(want))
    s.close()


def test_png_errors():
    s = _scene_with_rect()
    good = write_png([bytes([1, 2, 3, 4, 5, 6])], 2, 1, 8, 2)
    for bad in (b"", b"not a png at all", good[:-20], good.replace(b"IDAT", b"IDAX")):
        with pytest.raises(host.HostError):
            s.add_texture_png(bad)
    interlaced = bytearray(good)
    interlaced[8 + 8 + 12] = 1  # IHDR's interlace byte (its CRC is not checked)
    with pytest.raises(host.HostError, match="interlaced"):
        s.add_texture_png(bytes(interlaced))
    assert s.add_texture_png(good) == 1
    s.close()


# ---------------------------------------------------------------- glTF writing

class Gltf:
    """A glTF document under construction: one buffer, views and accessors appended as data arrives."""

    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "buffers": [{}], "bufferViews": [], "accessors": [], "meshes": [],
                    "nodes": [], "scenes": [{"nodes": []}], "scene": 0, "materials": [], "textures": [], "images": [],
                    "samplers": []}

    def view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.bin += data
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, kind, ctype, normalized=False):
        a = np.ascontiguousarray(array)
        acc = {"bufferView": self.view(a.tobytes()), "componentType": ctype, "count": len(a), "type": kind}
        if normalized:
            acc["normalized"] = True
        if kind == "VEC3" and ctype == 5126:
            acc["min"], acc["max"] = a.min(0).tolist(), a.max(0).tolist()
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def finish(self, path, mode):
        doc = json.loads(json.dumps(self.doc))
        doc = {k: v for k, v in doc.items() if v != []}
        doc["buffers"][0]["byteLength"] = len(self.bin)
        if mode == "glb":
            js = json.dumps(doc).encode()
            js += b" " * (-len(js) % 4)
            binc = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
            body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binc), 0x004E4942) + binc
            with open(path, "wb") as f:
                f.write(b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body)
            return
        if mode == "embedded":
            doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()
        else:
            name = os.path.basename(path) + " data.bin"  # a name that needs percent-encoding
            with open(os.path.join(os.path.dirname(path), name), "wb") as f:
                f.write(bytes(self.bin))
            doc["buffers"][0]["uri"] = name.replace(" ", "%20")
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)


def quat(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return [*(axis * np.sin(angle / 2)), float(np.cos(angle / 2))]


def mat_of_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trs(t, q, s):
    m = np.eye(4)
    m[:3, :3] = mat_of_quat(q) * np.asarray(s, float)[None, :]
    m[:3, 3] = t
    return m


def matrix_of_transform(t):
    """T · (Rx · Ry · Rz) · S of a Transform record, in doubles."""
    x, y, z = [float(v) for v in t["rotation"][:3]]
    rx = np.array([[1, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    rz = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = (rx @ ry @ rz) * np.asarray(t["scale"][:3], float)[None, :]
    m[:3, 3] = t["translation"][:3]
    return m


QUAD_POS = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], "<f4")
QUAD_NRM = np.array([[0, 0, 1]] * 4, "<f4")
QUAD_UV = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], "<f4")
QUAD_IDX = np.array([0, 1, 2, 0, 2, 3], "<u2")
TRI_POS = np.array([[0, 0, 0], [2, 0, 0], [0, 3, 0], [0, 0, 1], [2, 0, 1], [0, 3, 1]], "<f4")  # two triangles, no indices


def build_document(tmp_path, mode):
    g = Gltf()
    d = g.doc
    tex_img = np.arange(4 * 3 * 4, dtype=np.uint32).reshape(3, 4, 4) * 5 % 256
    png = write_png([tex_img[y].astype("u1").tobytes() for y in range(3)], 4, 3, 8, 6)
    d["images"] = [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()},
                   {"bufferView": g.view(png), "mimeType": "image/png"},
                   {"uri": "data:image/jpeg;base64," + base64.b64encode(b"\xff\xd8\xff\xe0 not really").decode()}]
    d["samplers"] = [{"magFilter": 9728}, {"magFilter": 9729, "wrapS": 33071}]
    d["textures"] = [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}, {"source": 2}]
    d["materials"] = [
        {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.5, 0.25, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.75,
                                  "baseColorTexture": {"index": 0}}},
        {"emissiveFactor": [1.0, 0.5, 0.25], "emissiveTexture": {"index": 1},
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}},
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 2}},
         "extensions": {"KHR_materials_ior": {"ior": 1.33}, "KHR_materials_transmission": {"transmissionFactor": 0.9}}},
    ]
    # mesh 0: an indexed quad with normals and uvs (u16 indices), plus a LINES primitive that must be skipped
    quad = {"attributes": {"POSITION": g.accessor(QUAD_POS, "VEC3", 5126), "NORMAL": g.accessor(QUAD_NRM, "VEC3", 5126),
                           "TEXCOORD_0": g.accessor(QUAD_UV, "VEC2", 5126)},
            "indices": g.accessor(QUAD_IDX, "SCALAR", 5123), "material": 0}
    lines = {"attributes": {"POSITION": quad["attributes"]["POSITION"]}, "mode": 1}
    # mesh 1: two triangles without indices or normals, positions interleaved with padding (byteStride 20),
    # uvs as normalised u8, no material
    inter = b"".join(p.tobytes() + b"\xAA" * 8 for p in TRI_POS)
    bv = g.view(inter, stride=20)
    d["accessors"].append({"bufferView": bv, "componentType": 5126, "count": 6, "type": "VEC3",
                           "min": TRI_POS.min(0).tolist(), "max": TRI_POS.max(0).tolist()})
    uv8 = np.array([[0, 0], [255, 0], [0, 255], [51, 51], [255, 102], [0, 204]], "u1")
    tri = {"attributes": {"POSITION": len(d["accessors"]) - 1, "TEXCOORD_0": g.accessor(uv8, "VEC2", 5121, normalized=True)}}
    # mesh 2: the quad again with u32 indices and u8 indices as two primitives, materials 1 and 2
    quad32 = dict(quad, indices=g.accessor(QUAD_IDX.astype("<u4"), "SCALAR", 5125), material=1)
    quad8 = dict(quad, indices=g.accessor(QUAD_IDX.astype("u1"), "SCALAR", 5121), material=2)
    d["meshes"] = [{"primitives": [quad, lines]}, {"primitives": [tri]}, {"primitives": [quad32, quad8]}]
    q1 = quat([1, 2, 3], 0.7)
    q2 = quat([0, 1, 0], np.pi / 2)  # sin b = 1: the gimbal case
    shear = (np.array([[1, 0.5, 0, 2], [0, 1, 0, 0], [0, 0, 1, -3], [0, 0, 0, 1]], float)).T.reshape(-1).tolist()  # column-major
    d["nodes"] = [
        {"name": "root", "translation": [1, 2, 3], "scale": [2, 2, 2], "children": [1, 2]},
        {"name": "quad A", "mesh": 0, "rotation": q1, "translation": [0, 0, 5]},
        {"name": "group", "rotation": q2, "children": [3]},
        {"name": "quad B (same mesh, mirrored)", "mesh": 0, "scale": [-1, 1.5, 1]},
        {"name": "triangles, sheared", "mesh": 1, "matrix": shear},
        {"name": "two primitives", "mesh": 2, "translation": [0, -4, 0]},
        {"name": "not in the scene", "mesh": 1},
    ]
    d["scenes"] = [{"nodes": [6]}, {"nodes": [0, 4, 5]}]
    d["scene"] = 1
    path = str(tmp_path / ("scene.glb" if mode == "glb" else "scene.gltf"))
    g.finish(path, mode)
    world = {
        "quad A": trs([1, 2, 3], [0, 0, 0, 1], [2, 2, 2]) @ trs([0, 0, 5], q1, [1, 1, 1]),
        "quad B": trs([1, 2, 3], [0, 0, 0, 1], [2, 2, 2]) @ trs([0, 0, 0], q2, [1, 1, 1]) @ trs([0, 0, 0], [0, 0, 0, 1], [-1, 1.5, 1]),
        "shear": np.array(shear).reshape(4, 4).T,
        "two": trs([0, -4, 0], [0, 0, 0, 1], [1, 1, 1]),
    }
    return path, world, tex_img


def srgb8_to_linear8(img):
    """IEC 61966-2-1 decoding of the three colour channels of an RGBA8 image to the nearest linear 8-bit value."""
    c = np.arange(256) / 255.0
    table = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    table = np.floor(table * 255.0 + 0.5).astype(np.uint32)
    out = np.asarray(img).astype(np.uint32).copy()
    out[..., :3] = table[out[..., :3]]
    return out


def geometry_of(arrays, g):
    go = arrays["geometry_offsets"]
    v0 = int(go["vertex"][g])
    v1 = int(go["vertex"][g + 1]) if g + 1 < len(go) else len(arrays["vertices"])
    i0 = int(go["index"][g])
    i1 = int(go["index"][g + 1]) if g + 1 < len(go) else len(arrays["indices"])
    return arrays["vertices"][v0:v1], arrays["indices"]["v"][i0:i1, :3]


def triangle_set(vertices, faces, matrix=None):
    """The triangles as a sorted array of corner positions (the BLAS build reorders faces)."""
    p = vertices["position"][:, :3].astype(np.float64)
    if matrix is not None:
        p = p @ matrix[:3, :3].T + matrix[:3, 3]
    t = p[faces].reshape(len(faces), 9)
    return t[np.lexsort(t.T[::-1])]


@pytest.mark.parametrize("mode", ["external", "embedded", "glb"])
def test_gltf_scene_graph_meshes_materials_textures(tmp_path, mode):
    path, world, tex_img = build_document(tmp_path, mode)
    s = host.Scene()
    info = s.load_gltf(path)
    assert info == {"n_meshes": 4, "n_instances": 5, "n_materials": 4, "n_textures": 2, "n_baked": 1,
                    "n_skipped_primitives": 1, "n_skipped_images": 1}
    arrays = s.bind()
    s.close()
    inst, xf = arrays["instances"], arrays["transforms"]
    assert len(inst) == 5 and len(arrays["geometry_offsets"]) == 4
    # spawn order: depth first from the scene's roots — quad A, quad B, sheared triangles, two primitives
    ga, gb, gs, g32, g8 = [int(i["geometry"]) for i in inst]
    assert ga == gb and len({ga, gs, g32, g8}) == 4  # the quad is one mesh shared by two nodes
    for k, name in ((0, "quad A"), (1, "quad B"), (3, "two"), (4, "two")):
        got = matrix_of_transform(xf[int(inst[k]["transform"])])
        assert np.allclose(got, world[name], atol=2e-6), name
    assert np.allclose(matrix_of_transform(xf[int(inst[2]["transform"])]), np.eye(4))  # the sheared one was baked
    # geometry: the file's positions untouched for shared meshes, moved by the world matrix for the baked one
    v, f = geometry_of(arrays, ga)
    assert np.array_equal(triangle_set(v, f), triangle_set({"position": np.c_[QUAD_POS, np.ones(4)]}, QUAD_IDX.reshape(2, 3).astype(int)))
    assert np.array_equal(v["normal"][:, :3], QUAD_NRM) and np.array_equal(v["uv"][:, :2], QUAD_UV)
    v, f = geometry_of(arrays, gs)
    want = triangle_set({"position": np.c_[TRI_POS, np.ones(6)]}, np.arange(6).reshape(2, 3), world["shear"])
    assert np.allclose(triangle_set(v, f), want, atol=1e-6)
    assert np.allclose(v["uv"][:, :2], np.array([[0, 0], [255, 0], [0, 255], [51, 51], [255, 102], [0, 204]]) / 255.0)
    n = v["normal"][:, :3]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6) and np.allclose(np.abs(n[:, 2]), 1.0, atol=1e-6)
    # materials
    m = arrays["materials"]
    ma, ms, m32, m8 = [m[int(i["material"])] for i in (inst[0], inst[2], inst[3], inst[4])]
    assert int(inst[0]["material"]) == int(inst[1]["material"])
    assert np.allclose(ma["colour"], [0.8, 0.5, 0.25, 1.0]) and np.isclose(ma["metallic"], 0.25) and np.isclose(ma["roughness"], 0.75)
    assert ma["colour_texture"] == 1 and ma["emissive_texture"] == 0
    assert np.allclose(ms["colour"], 1.0) and ms["metallic"] == 1.0 and ms["roughness"] == 1.0 and ms["colour_texture"] == 0
    assert np.allclose(m32["emissive"][:3], [4.0, 2.0, 1.0]) and m32["emissive_texture"] == 2
    assert np.isclose(m8["ior"], 1.33) and np.isclose(m8["transmission"], 0.9) and m8["colour_texture"] == 0  # a JPEG: skipped
    # textures: the same image twice, the first with the NEAREST sampler
    tex = arrays["textures"]
    assert [int(t["filter"]) for t in tex] == [ffi.FILTER_NEAREST, ffi.FILTER_BILINEAR]
    # base-colour and emissive images are sRGB-encoded in glTF: the store holds the decoded (linear) colour
    # channels, alpha as it is
    lin = srgb8_to_linear8(tex_img)
    assert lin[..., :3].max() < tex_img[..., :3].max() and np.array_equal(lin[..., 3], tex_img[..., 3])
    want = rgba(lin[..., 0], lin[..., 1], lin[..., 2], lin[..., 3]).reshape(-1)
    assert np.array_equal(arrays["texels"], np.concatenate([want, want]))
    assert tuple(tex[0][["width", "height"]]) == (4, 3)
    # light sources: the instance with the emissive material
    assert list(arrays["light_sources"]) == [3]


def test_gltf_three_containers_give_the_same_scene(tmp_path):
    out = []
    for mode in ("external", "embedded", "glb"):
        d = tmp_path / mode
        d.mkdir()
        path, _, _ = build_document(d, mode)
        s = host.Scene()
        s.load_gltf(path)
        out.append(s.bind())
        s.close()
    for k in out[0]:
        assert out[0][k].tobytes() == out[1][k].tobytes() == out[2][k].tobytes(), k


def test_gltf_without_scenes_uses_the_root_nodes_and_errors_are_reported(tmp_path):
    g = Gltf()
    g.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": g.accessor(QUAD_POS, "VEC3", 5126)},
                                        "indices": g.accessor(QUAD_IDX, "SCALAR", 5123)}]}]
    g.doc["nodes"] = [{"children": [1]}, {"mesh": 0, "translation": [0, 0, 2]}, {"mesh": 0}]
    g.doc["scenes"] = []
    del g.doc["scene"]
    path = str(tmp_path / "roots.gltf")
    g.finish(path, "embedded")
    s = host.Scene()
    assert s.load_gltf(path)["n_instances"] == 2
    arrays = s.bind()
    assert sorted(float(t["translation"][2]) for t in arrays["transforms"][arrays["instances"]["transform"]]) == [0.0, 2.0]
    # computed normals for a quad in the z = 0 plane, counter-clockwise
    assert np.allclose(arrays["vertices"]["normal"][:4, :3], [[0, 0, 1]] * 4)
    s.close()

    def fails(text, match):
        p = tmp_path / "bad.gltf"
        p.write_text(text)
        s = host.Scene()
        with pytest.raises(host.HostError, match=match):
            s.load_gltf(str(p))
        s.close()

    doc = json.load(open(path))
    fails("{ not json", "JSON")
    fails(json.dumps(dict(doc, asset={"version": "1.0"})), "version")
    fails(json.dumps(dict(doc, nodes=[{"mesh": 7}])), "no triangle primitive")
    bad = json.loads(json.dumps(doc))
    bad["accessors"][1]["count"] = 600  # indices past their view
    fails(json.dumps(bad), "no triangle primitive")
    bad = json.loads(json.dumps(doc))
    bad["bufferViews"][0]["byteLength"] = 10 ** 9
    fails(json.dumps(bad), "no triangle primitive")
    s = host.Scene()
    with pytest.raises(host.HostError, match="cannot open"):
        s.load_gltf(str(tmp_path / "missing.gltf"))
    s.close()


def test_gltf_node_graph_that_is_no_tree_is_walked_once(tmp_path):
    """glTF demands a strict tree.  A node that lists itself, a cycle through several nodes and a diamond forty
    levels deep (2^40 paths) each load at once, every node instanced the first time it is reached; a child index
    that is no finite integer is ignored."""
    import time
    g = Gltf()
    g.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": g.accessor(QUAD_POS, "VEC3", 5126)},
                                        "indices": g.accessor(QUAD_IDX, "SCALAR", 5123)}]}]

    def load(nodes, roots, name, patch=None):
        g.doc["nodes"] = nodes
        g.doc["scenes"] = [{"nodes": roots}]
        path = str(tmp_path / name)
        g.finish(path, "embedded")
        if patch:
            text = open(path).read()
            assert patch[0] in text
            open(path, "w").write(text.replace(*patch))
        s = host.Scene()
        t0 = time.perf_counter()
        info = s.load_gltf(path)
        assert time.perf_counter() - t0 < 5.0
        s.close()
        return info["n_instances"]

    assert load([{"mesh": 0, "children": [0, 0]}], [0], "self.gltf") == 1
    assert load([{"mesh": 0, "children": [1]}, {"mesh": 0, "children": [2, 0]}, {"mesh": 0, "children": [0, 1]}], [0], "cycle.gltf") == 3
    diamond = [{"mesh": 0, "children": [i + 1, i + 1]} for i in range(40)] + [{"mesh": 0}]
    assert load(diamond, [0], "diamond.gltf") == 41
    assert load([{"mesh": 0, "children": [1, 987654321]}, {"mesh": 0}], [0], "huge.gltf", ("987654321", "1e999")) == 2


def test_gltf_baked_mesh_keeps_the_files_normals(tmp_path):
    """A sheared node bakes its matrix into a copy of the mesh: positions by the matrix, the file's normals by its
    inverse transpose (renormalised) — not re-derived from the faces."""
    g = Gltf()
    tilted = np.array([[0.6, 0.0, 0.8]] * 4, "<f4")  # authored normals that no face of the flat quad would give
    g.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": g.accessor(QUAD_POS, "VEC3", 5126),
                                                       "NORMAL": g.accessor(tilted, "VEC3", 5126)},
                                        "indices": g.accessor(QUAD_IDX, "SCALAR", 5123)}]}]
    m = np.array([[1, 0.5, 0.25, 2], [0, 1.5, 0, 0], [0.3, 0, 1, -3], [0, 0, 0, 1]], float)
    mirror = m @ np.diag([-1.0, 1, 1, 1])
    g.doc["nodes"] = [{"mesh": 0, "matrix": m.T.reshape(-1).tolist()}, {"mesh": 0, "matrix": mirror.T.reshape(-1).tolist()}]
    g.doc["scenes"] = [{"nodes": [0, 1]}]
    path = str(tmp_path / "sheared.gltf")
    g.finish(path, "embedded")
    s = host.Scene()
    assert s.load_gltf(path)["n_baked"] == 2
    arrays = s.bind()
    s.close()
    for k, mat in enumerate((m, mirror)):
        v, _ = geometry_of(arrays, int(arrays["instances"][k]["geometry"]))
        want = np.linalg.inv(mat[:3, :3]).T @ np.array([0.6, 0.0, 0.8])
        want /= np.linalg.norm(want)
        assert np.allclose(v["normal"][:, :3], want, atol=1e-6), k
        assert np.allclose(v["position"][:, :3], QUAD_POS @ mat[:3, :3].T + mat[:3, 3], atol=1e-6)


@pytest.mark.gpu
def test_gltf_scene_renders_on_the_device_as_in_the_oracle(tmp_path):
    """The loaded scene end to end: instanced and baked meshes, textured and emissive materials, through
    the uber kernels and the per-material kernels with light sampling — radiance sums equal to the oracle's."""
    from oracle import pyoracle as po
    from raytracer_amd import pathtracer
    path, _, _ = build_document(tmp_path, "glb")
    s = host.Scene()
    s.load_gltf(path)
    arrays = s.bind()
    s.close()
    W, H = 80, 60
    src = host.make_sample_sources(W, H, 31)
    cam = host.default_camera(W, H)
    cam["position"][0] = (1.0, 0.0, -9.0)
    with pathtracer.Pathtracer(0) as pt:
        pt.upload_scene(arrays)
        sc = po.Scene(arrays)
        for mode, nee in ((0, 0), (1, 1)):
            kw = dict(max_bounces=4, seed=31, shade_mode=mode, nee=nee)
            pt.create_state(W, H, 2000, src, pathtracer.default_params(**kw))
            pt.set_camera(cam)
            pt.render_spp(3)
            s4, cnt = pt.read_accum()
            w4, wcnt, _ = sc.render(cam, src, W, H, po.default_params(**kw), 0, 3)
            assert np.array_equal(cnt, wcnt) and np.array_equal(s4, w4)
        rays = np.zeros(W * H, ffi.Ray)
        rays["pos"] = cam["position"][0]
        ys, xs = np.mgrid[0:H, 0:W]
        d = np.stack([(xs + 0.5) / W - 0.5, 0.5 - (ys + 0.5) / H, np.ones_like(xs, float)], -1).reshape(-1, 3)
        rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
        rays["last_inst"] = rays["last_prim"] = 0x7FFFFFFF
        got, want = pt.trace_closest(rays), sc.trace_closest(rays)
        assert got.tobytes() == want.tobytes()
        # the instanced quad, the baked triangles and the two-primitive quad are all somebody's first hit from here
        assert set(want["instance_id"][want["instance_id"] != ffi.NO_HIT].tolist()) >= {0, 2, 3}
