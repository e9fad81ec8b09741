#!/bin/bash
# Tuning aid: build variants of libwf_lde.so into build/exp_<name>/ -- any flags (a kernel change behind a macro of its own), or
# the diagnostic switches of the experiment build (-DWF_EXPERIMENTS -DWF_EXP_SKIP_LOAD / _SKIP_NTT / _SKIP_STORE / _LOCAL_STORE=1|2 /
# _STAMPS, -DWF_EXP_R8_NO_SPLIT = the f128 round order 8, 8, 8, 2; the product build ignores them) -- and time cfg 2 with each
# (run the timing part on the GPU box:  WF_LDE_LIB=build/exp_<name>/libwf_lde.so python scripts/time_config.py 1 1 20 3 8 1).
#   scripts/exp_variants.sh [name "flags"]...   (default: the time-attribution set of DESIGN.md §4)
set -e
cd "$(dirname "$0")/.."
build_variant() {  # name, extra flags
    rm -rf build/exp_$1
    mkdir -p build/exp_$1/pkg
    cp -r include build/exp_$1/include
    cp -r starkpack-winterfell_amd/csrc build/exp_$1/pkg/csrc
    rm -rf build/exp_$1/pkg/csrc/obj build/exp_$1/pkg/csrc/*.so
    make -s -j8 -C build/exp_$1/pkg/csrc libwf_lde.so \
        HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed -ffp-contract=off $2"
    cp build/exp_$1/pkg/csrc/libwf_lde.so build/exp_$1/libwf_lde.so
    rm -rf build/exp_$1/pkg build/exp_$1/include
}
if [ $# -ge 2 ]; then
    while [ $# -ge 2 ]; do build_variant "$1" "$2"; shift 2; done
    exit 0
fi
build_variant skip_ntt "-DWF_EXPERIMENTS -DWF_EXP_SKIP_NTT"
build_variant skip_store "-DWF_EXPERIMENTS -DWF_EXP_SKIP_STORE"
build_variant skip_load "-DWF_EXPERIMENTS -DWF_EXP_SKIP_LOAD"
build_variant skip_mem "-DWF_EXPERIMENTS -DWF_EXP_SKIP_LOAD -DWF_EXP_SKIP_STORE"
build_variant skip_ntt_store "-DWF_EXPERIMENTS -DWF_EXP_SKIP_NTT -DWF_EXP_SKIP_STORE"
build_variant skip_ntt_load "-DWF_EXPERIMENTS -DWF_EXP_SKIP_NTT -DWF_EXP_SKIP_LOAD"
