// UNVERIFIED (no Rust toolchain where this was written).
// VRM_LIB_DIR = directory holding libvrm.so (built by `make -C volume_renderer_amd/csrc`).
fn main() {
    let dir = std::env::var("VRM_LIB_DIR").expect("set VRM_LIB_DIR to the directory that holds libvrm.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=vrm");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=VRM_LIB_DIR");
}
