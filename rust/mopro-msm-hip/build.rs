// Link against the in-tree libmsm_hip.so (built by `make -C gpu-acceleration_amd/csrc`).
fn main() {
    let dir = std::env::var("MSM_HIP_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../../gpu-acceleration_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=msm_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=MSM_HIP_LIB_DIR");
}
