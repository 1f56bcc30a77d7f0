// Link against libblsgpu.so (built by `python -c "import __graft_entry__ as g; g.build()"` at the repository root).
// BLSGPU_LIB_DIR overrides the search path; the default is ../../bls12_381_amd relative to this crate.
fn main() {
    let dir = std::env::var("BLSGPU_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../bls12_381_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=blsgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=BLSGPU_LIB_DIR");
}
