"""The boundary, locked from four sides without a GPU and without rustc: include/mirt.h (the
contract), libmirt.so (what is exported), raytracer_amd/ffi.py (what the harness binds) and
rust_shim/ (what a maintainer of the reference would bind) must agree."""
import os
import re
import subprocess

from conftest import ROOT
from raytracer_amd import ffi


def strip_c_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def header_functions(path):
    """{name: number of arguments} of every function a header declares."""
    text = strip_c_comments(open(path).read())
    out = {}
    for m in re.finditer(r"\b(mirt[h]?_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def header_struct_fields(path, name):
    text = strip_c_comments(open(path).read())
    body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.search(r"(\w+)\s*(\[\w*\])*$", part.strip()).group(1))
    return fields


def rust_functions(path):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def rust_struct_fields(path, name):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    body = re.search(r"pub struct %s\s*\{(.*?)\}" % name, text, flags=re.S).group(1)
    return re.findall(r"pub (\w+)\s*:", body)


def test_header_library_harness_and_rust_shim_agree():
    want = header_functions(os.path.join(ROOT, "include", "mirt.h"))
    assert len(want) >= 48 and want["mirt_state_create"] == 7 and want["mirt_default_params"] == 1
    # the harness binds every entry point, and nothing the header does not have
    assert sorted(ffi.DEVICE_SYMBOLS) == sorted(want)
    # the library exports every one of them (dynamic symbol table; no GPU needed)
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ffi.LIB_DIR, "libmirt.so")],
                        capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mirt_\w+)", nm))
    assert set(want) <= exported, sorted(set(want) - exported)
    # the Rust declarations: same functions, same number of arguments
    rust = rust_functions(os.path.join(ROOT, "rust_shim", "src", "ffi.rs"))
    assert rust == want
    # the structs that cross by pointer: same fields in the same order
    header = os.path.join(ROOT, "include", "mirt.h")
    shim = os.path.join(ROOT, "rust_shim", "src", "ffi.rs")
    for name in ("MirtSceneDesc", "MirtParams", "MirtTexture", "MirtRay", "MirtHitId", "MirtTraversalCounts",
                 "MirtStats", "MirtClassicMaterialRef"):
        assert rust_struct_fields(shim, name) == header_struct_fields(header, name), name
    # ... and the ctypes mirrors
    assert [f for f, _ in ffi.SceneDesc._fields_] == header_struct_fields(header, "MirtSceneDesc")
    assert [f for f, _ in ffi.Params._fields_] == header_struct_fields(header, "MirtParams")
    assert [f for f, _ in ffi.Stats._fields_] == header_struct_fields(header, "MirtStats")
    assert list(ffi.Texture.names) == header_struct_fields(header, "MirtTexture")


def test_host_library_exports_its_header():
    want = header_functions(os.path.join(ROOT, "include", "mirt_host.h"))
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ffi.LIB_DIR, "libmirt_host.so")],
                        capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mirth_\w+)", nm))
    assert len(want) > 20 and set(want) <= exported, sorted(set(want) - exported)


def test_rust_build_script_compiles_every_device_source():
    build_rs = open(os.path.join(ROOT, "rust_shim", "build.rs")).read()
    listed = set(re.findall(r'"(\w+\.hip)"', build_rs))
    present = {f for f in os.listdir(os.path.join(ROOT, "raytracer_amd", "csrc", "device")) if f.endswith(".hip")}
    assert listed == present
    # the flags of the arithmetic contract are the Makefile's
    makefile = open(os.path.join(ROOT, "Makefile")).read()
    for flag in ("-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "--offload-arch=gfx950", "-fno-gpu-rdc"):
        assert flag in build_rs and flag.replace("gfx950", "$(ARCH)") in makefile.replace("$(ARCH)", "$(ARCH)") or flag in makefile
    # every ffi call the shim's systems make is declared in ffi.rs
    rust = rust_functions(os.path.join(ROOT, "rust_shim", "src", "ffi.rs"))
    for name in os.listdir(os.path.join(ROOT, "rust_shim", "src")):
        for call in re.findall(r"ffi::(mirt_\w+)\(", open(os.path.join(ROOT, "rust_shim", "src", name)).read()):
            assert call in rust, (name, call)
