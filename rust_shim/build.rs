// build.rs — replaces the slangc loop of the reference's build.rs:43-58: the
// wavefront kernels are compiled by hipcc for gfx950 into a static library that
// the crate links.  MIRT_DIR points at a checkout of the MI355X library
// (default: ../raytracer_mi355x).
use std::{path::PathBuf, process::Command};

// every translation unit of raytracer_amd/csrc/device/
const DEVICE_SOURCES: [&str; 5] = ["api.hip", "kernels.hip", "classic.hip", "build.hip", "diag.hip"];

fn main() {
    let out = PathBuf::from(std::env::var("OUT_DIR").unwrap());
    let root = PathBuf::from(std::env::var("MIRT_DIR").unwrap_or_else(|_| "../raytracer_mi355x".into()));
    let src = root.join("raytracer_amd/csrc/device");
    let hipcc = std::env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    println!("cargo:rerun-if-changed={}", src.display());
    println!("cargo:rerun-if-changed={}", root.join("include/mirt.h").display());

    let mut objects = vec![];
    for name in DEVICE_SOURCES {
        let obj = out.join(format!("{name}.o"));
        let status = Command::new(&hipcc)
            .args([
                "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC",
                // the arithmetic contract shared with the CPU checker: no implicit
                // contraction, correctly rounded division and square root
                "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc",
            ])
            .arg(format!("-I{}", root.join("include").display()))
            .arg("-c")
            .arg(src.join(name))
            .arg("-o")
            .arg(&obj)
            .status()
            .expect("failed to run hipcc");
        assert!(status.success(), "hipcc failed on {name}");
        objects.push(obj);
    }
    let lib = out.join("libmirt.a");
    let _ = std::fs::remove_file(&lib);
    let status = Command::new("ar").arg("rcs").arg(&lib).args(&objects).status().expect("failed to run ar");
    assert!(status.success(), "ar failed");

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=static=mirt");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=stdc++");
}
