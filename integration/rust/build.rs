// UNVERIFIED (no Rust toolchain in the build image).  Replaces the reference's build.rs:1-11.
use std::env;

fn main() -> anyhow::Result<()> {
    let target = env::var("CARGO_CFG_TARGET_OS")?;
    if target == "windows" {
        let mut res = winresource::WindowsResource::new();
        res.set_icon("./res/icon.ico");
        res.compile()?;
    }
    // libtrayracer_hip.a: `make -C raytracing_amd/csrc` (hipcc --offload-arch=gfx950 -ffp-contract=off)
    if let Ok(dir) = env::var("TRAYRACER_HIP_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=static=trayracer_hip");
        println!("cargo:rustc-link-search=native=/opt/rocm/lib");
        println!("cargo:rustc-link-lib=dylib=amdhip64");
        println!("cargo:rustc-link-lib=dylib=stdc++");
        println!("cargo:rustc-cfg=trayracer_hip");
    }
    println!("cargo:rerun-if-env-changed=TRAYRACER_HIP_DIR");
    Ok(())
}
