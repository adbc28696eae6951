// libfwgpu.so is built outside cargo (hipcc --offload-arch=gfx950, firewheel_amd/csrc/Makefile of the fwgpu repository).
// FWGPU_LIB_DIR names the directory that holds it; it needs libamdhip64 (ROCm) at run time, nothing else.
fn main() {
    println!("cargo:rerun-if-env-changed=FWGPU_LIB_DIR");
    if let Ok(dir) = std::env::var("FWGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    // FWGPU_NO_LINK=1: tests that never call into the library (tests/reference_digests.rs: the parity scenarios on the real
    // firewheel-graph, scripts/pin_parity.sh) build and run on a machine without ROCm
    println!("cargo:rerun-if-env-changed=FWGPU_NO_LINK");
    if std::env::var("FWGPU_NO_LINK").map(|v| v == "1").unwrap_or(false) {
        return;
    }
    println!("cargo:rustc-link-lib=dylib=fwgpu");
}
