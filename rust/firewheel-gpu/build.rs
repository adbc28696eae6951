// libfwgpu.so is built outside cargo (hipcc --offload-arch=gfx950, firewheel_amd/csrc/Makefile of the fwgpu repository).
// FWGPU_LIB_DIR names the directory that holds it; it needs libamdhip64 (ROCm) at run time, nothing else.
fn main() {
    println!("cargo:rerun-if-env-changed=FWGPU_LIB_DIR");
    if let Ok(dir) = std::env::var("FWGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=fwgpu");
}
