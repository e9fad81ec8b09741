"""starkpack-winterfell_amd: MI355X (gfx950) implementation of the winter-prover hot path -- trace / constraint
low-degree extension and BLAKE3 Merkle commitment -- behind the C ABI of include/wf_lde.h.

    capi   : ctypes binding of libwf_lde.so (the drop-in boundary)
    build  : compiles csrc/ with hipcc for gfx950
    shard  : multi-GPU layout (independent proofs per GPU + one all-gather of roots)

There is no CPU fallback in this package: without libwf_lde.so and a HIP device every compute call raises.
"""
__all__ = ["capi", "build", "shard"]
