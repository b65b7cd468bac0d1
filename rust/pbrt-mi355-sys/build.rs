// Links libpbrt_mi355.so.  PBRT_MI355_LIB_DIR names the directory that holds it (default: the in-tree build output
// pbrt-rs_amd/ two levels above this crate); the library itself links the ROCm runtime (libamdhip64).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("PBRT_MI355_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../pbrt-rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=pbrt_mi355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PBRT_MI355_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/pbrt_mi355.h");
}
