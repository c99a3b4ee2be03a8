// Links libfmd_hip.so.  Set FMD_LIB_DIR to the directory that holds it (rtl-sdr-rs_amd/ in this repository).
fn main() {
    if let Ok(dir) = std::env::var("FMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=fmd_hip");
    println!("cargo:rerun-if-env-changed=FMD_LIB_DIR");
}
