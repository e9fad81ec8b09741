// Points the linker at libwf_lde.so.  WF_LDE_LIB_DIR = the directory that holds it
// (starkpack-winterfell_amd/csrc after `make -C starkpack-winterfell_amd/csrc`).
fn main() {
    let dir = std::env::var("WF_LDE_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=wf_lde");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=WF_LDE_LIB_DIR");
}
