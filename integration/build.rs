// build.rs of the reference crate with the MI355X provider enabled (feature "hip"): link libspartan_hip.so, built by
// `python -c 'import __graft_entry__ as g; g.build()'` in this repository (spartan2_amd/lib/). SPARTAN_HIP_LIB_DIR names that directory.
fn main() {
  if std::env::var("CARGO_FEATURE_HIP").is_ok() {
    let dir = std::env::var("SPARTAN_HIP_LIB_DIR").expect("SPARTAN_HIP_LIB_DIR = <repo>/spartan2_amd/lib");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=spartan_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SPARTAN_HIP_LIB_DIR");
  }
}
