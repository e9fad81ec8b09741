// Bring-up microbenchmark: issue rate of individual gfx950 VALU instructions (wave-instructions per SIMD cycle).
//   hipcc -O3 --offload-arch=gfx950 scripts/instr_rate.hip -o build/instr_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                        \
    __global__ void name(uint32_t *out, int iters) {                              \
        uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;         \
        uint64_t p = a, q = b;                                                    \
        (void)p; (void)q;                                                         \
        for (int i = 0; i < iters; i++) { REP64(body) }                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)p ^ (uint32_t)q; \
    }

KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a)); asm volatile("v_alignbit_b32 %0, %0, %0, 12" : "+v"(c));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p) : "v"(b), "v"(d) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(d), "v"(b) : "vcc");)
KERNEL(k_lshl_add64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p) : "v"(q)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(p));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d) : "vcc");)
KERNEL(k_addco, asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a), "+v"(c) : "v"(b), "v"(d) : "vcc"); )
KERNEL(k_cmp64, asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p), "v"(q) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q), "v"(p) : "vcc");)
KERNEL(k_fma32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q) : "v"(p));)
// round 3: candidates for cheaper BLAKE3 rotations / field shifts
KERNEL(k_xor_sdwa, asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(c) : "v"(d));)
KERNEL(k_xor_sdwa_keep, asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(a)); asm volatile("v_lshlrev_b32 %0, 9, %0" : "+v"(c));)
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a) : "v"(b)); asm volatile("v_lshl_or_b32 %0, %0, 9, %1" : "+v"(c) : "v"(d));)
KERNEL(k_lshl64, asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(p)); asm volatile("v_lshlrev_b64 %0, 9, %0" : "+v"(q));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(d));)
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d));)
KERNEL(k_add_dpp, asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d));)
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_addco_only, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc"); asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(c) : "v"(d) : "vcc");)
KERNEL(k_pk_add16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c) : "v"(d));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));)
KERNEL(k_mad64_const0, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(b), "v"(d) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(d), "v"(b) : "vcc");)

// round 4: the conditional move with its mask in an SGPR pair that nothing in the loop writes (the round-3 line above clobbers VCC in
// every statement, so hipcc puts an s_nop between the statements and the figure is not the instruction's); compare + select as the
// field code issues them; carry-in additions alone; and the double-precision pipe (a 104-bit product from two FMAs is the candidate
// for a cheaper f128 product: docs/STATUS.md section 7c)
__global__ void k_cndmask_sgpr(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
    const uint64_t mask = __builtin_amdgcn_read_exec() ^ (0x5555555555555555ull * (uint64_t)(iters & 1));
    for (int i = 0; i < iters; i++) {
        REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(mask)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(c) : "v"(d), "s"(mask));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(k_addc_only, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c) : "v"(d) : "vcc");)
#define KERNEL_F64(name, body)                                                    \
    __global__ void name(uint32_t *out, int iters) {                              \
        double x = threadIdx.x + 1.5, y = 1.0000001, z = 0.75, w = 1.25;           \
        for (int i = 0; i < iters; i++) { REP64(body) }                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x + z);           \
    }
KERNEL_F64(k_fma64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(w)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z) : "v"(w), "v"(y));)
KERNEL_F64(k_mul64f, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(z) : "v"(w));)
KERNEL_F64(k_add64f, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(z) : "v"(w));)
__global__ void k_cvt(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x + 3, c = threadIdx.x ^ 0x55;
    double x, z;
    for (int i = 0; i < iters; i++) {
        REP64(asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x) : "v"(a)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a) : "v"(x)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(z) : "v"(c)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(c) : "v"(z));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ c;
}

template <class K>
static void run(const char *name, K kern, uint32_t *out) {
    const int blocks = 256 * 8, threads = 256, iters = 200;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * (threads / 64) * iters * 64 * 2;  // 2 instructions per REP body
    const double per_simd_per_s = wave_instr / (256.0 * 4) / (ms * 1e-3);
    printf("%-14s %8.3f ms  %6.2f G wave-instr/s per SIMD  (= %4.2f cycles per wave-instr at 2.4 GHz)\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    uint32_t *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run("v_add_u32", k_add, out);
    run("v_xor_b32", k_xor, out);
    run("v_alignbit", k_alignbit, out);
    run("v_add3_u32", k_add3, out);
    run("v_mul_lo_u32", k_mul_lo, out);
    run("v_mul_hi_u32", k_mul_hi, out);
    run("v_mad_u64_u32", k_mad64, out);
    run("v_lshl_add_u64", k_lshl_add64, out);
    run("v_cndmask", k_cndmask, out);
    run("add_co+addc", k_addco, out);
    run("v_cmp_lt_u64", k_cmp64, out);
    run("v_fma_f32", k_fma32, out);
    run("v_mad_u32_u24", k_mad24, out);
    run("v_pk_fma_f32", k_pkfma, out);
    run("xor_sdwa_pad", k_xor_sdwa, out);
    run("xor_sdwa_keep", k_xor_sdwa_keep, out);
    run("v_perm_b32", k_perm, out);
    run("v_xad_u32", k_xad, out);
    run("v_lshlrev_b32", k_lshl, out);
    run("v_lshl_or_b32", k_lshl_or, out);
    run("v_lshlrev_b64", k_lshl64, out);
    run("v_mov_b32", k_mov, out);
    run("v_mov_dpp", k_mov_dpp, out);
    run("v_add_dpp", k_add_dpp, out);
    run("v_mul_u32_u24", k_mul24, out);
    run("v_add_co only", k_addco_only, out);
    run("v_pk_add_u16", k_pk_add16, out);
    run("v_bfi_b32", k_bfi, out);
    run("mad64 (+0)", k_mad64_const0, out);
    run("cndmask sgpr", k_cndmask_sgpr, out);
    run("cmp+cndmask/2", k_cmp_cnd, out);  // (one statement of two instructions per REP body: halve the cycles for the pair... the line is per instruction of a 2-instruction body)
    run("v_addc_co only", k_addc_only, out);
    run("v_fma_f64", k_fma64, out);
    run("v_mul_f64", k_mul64f, out);
    run("v_add_f64", k_add64f, out);
    run("cvt u32<->f64 x2", k_cvt, out);  // (four instructions per REP body: double the cycles)
    return 0;
}
