// UNVERIFIED (no Rust toolchain in the build image).  Lines to ADD inside `fn main()` of the reference's build.rs
// (build.rs:1-11, which only embeds the Windows icon), before its final `Ok(())`.
//
// libtrayracer_hip.a comes from `make -C raytracing_amd/csrc` (hipcc --offload-arch=gfx950 -ffp-contract=off); without
// TRAYRACER_HIP_DIR the crate builds as before and `cfg(trayracer_hip)` stays off.
fn link_trayracer_hip() {
    println!("cargo:rerun-if-env-changed=TRAYRACER_HIP_DIR");
    if let Ok(dir) = std::env::var("TRAYRACER_HIP_DIR") {
        for line in [
            format!("cargo:rustc-link-search=native={dir}"),
            "cargo:rustc-link-lib=static=trayracer_hip".to_owned(),
            "cargo:rustc-link-search=native=/opt/rocm/lib".to_owned(),
            "cargo:rustc-link-lib=dylib=amdhip64".to_owned(),
            "cargo:rustc-link-lib=dylib=stdc++".to_owned(),
            "cargo:rustc-cfg=trayracer_hip".to_owned(),
        ] {
            println!("{line}");
        }
    }
}
