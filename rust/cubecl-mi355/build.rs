// Links the prebuilt C-ABI library (cubecl_amd/csrc/libmi355cube.so, built by `make` with hipcc
// for gfx950).  MI355CUBE_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("MI355CUBE_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../cubecl_amd/csrc", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mi355cube");
    println!("cargo:rerun-if-env-changed=MI355CUBE_LIB_DIR");
}
