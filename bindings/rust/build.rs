// Links libxhe.so (the MI355X batch-verification library) when the crate is built with `--features mi355x`.
// XHE_LIB_DIR names the directory that holds it.
fn main() {
    println!("cargo:rerun-if-env-changed=XHE_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_MI355X").is_some() {
        if let Some(dir) = std::env::var_os("XHE_LIB_DIR") {
            println!("cargo:rustc-link-search=native={}", dir.to_string_lossy());
        }
        println!("cargo:rustc-link-lib=dylib=xhe");
    }
}
