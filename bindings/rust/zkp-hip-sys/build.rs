// Locates libzkp_hip.so (and, next to it, libzkp_hip_lat.so, which the main library dlopens by itself for calls of a few proofs).
//   ZKP_HIP_LIB_DIR   directory that holds libzkp_hip.so          (default: <repo>/zk-paillier_amd, relative to this crate)
//   ROCM_PATH         ROCm install whose runtime the library needs   (default: /opt/rocm)
// The library is built by `python -c "import __graft_entry__ as g; g.build()"` at the repository root (hipcc, --offload-arch=gfx950).
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR"));
    let default_dir = manifest.join("..").join("..").join("..").join("zk-paillier_amd");
    let lib_dir = env::var("ZKP_HIP_LIB_DIR").map(PathBuf::from).unwrap_or(default_dir);
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    if !lib_dir.join("libzkp_hip.so").exists() {
        panic!(
            "libzkp_hip.so not found in {} (set ZKP_HIP_LIB_DIR; there is no CPU fallback to link instead)",
            lib_dir.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=zkp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    println!("cargo:rerun-if-env-changed=ZKP_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=build.rs");
}
