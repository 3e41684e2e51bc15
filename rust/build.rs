// Links the MI355X k-mer counter (libkmc.so, built by `make -C ../k-mer-count_amd`).
fn main() {
    let dir = std::env::var("KMC_LIB_DIR").unwrap_or_else(|_| "../k-mer-count_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=kmc");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=KMC_LIB_DIR");
}
