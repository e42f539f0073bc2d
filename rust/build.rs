// build.rs for thomasantony/splat once it binds libsplat_hip.so (include/splat_hip.h).
// UNTESTED here (no cargo in the authoring image).  libsplat_hip.so is built by
// `make -C splat_amd/csrc` (hipcc --offload-arch=gfx950); SPLAT_HIP_DIR is the directory holding it.
fn main() {
    let dir = std::env::var("SPLAT_HIP_DIR").expect("set SPLAT_HIP_DIR to the directory holding libsplat_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=splat_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=SPLAT_HIP_DIR");
}
