// Constraint side of the path from the combined constraint EVALUATIONS to the resident constraint commitment without
// the composition polynomial visiting the host (SURVEY.md §8 rows a19-a21 chained):
//   ConstraintEvaluationTable::into_comb_poly      /root/reference/prover/src/constraints/evaluation_table.rs:166-186
//       (its tail: fft::interpolate_poly_with_offset over the constraint evaluation domain; the division by the
//        divisors in front of it is AIR-specific and stays with the caller)
//   the STARKPack combination over the packed traces   prover/src/lib.rs:442-453
//       final = comb_0 + sum_{i >= 1} comb_i * final_coeff^i
//   CompositionPoly::new / segment                  prover/src/constraints/composition_poly.rs:21-41, 86-98
//       (column c = coefficients [c * trace_length, (c + 1) * trace_length): the polynomial buffer IS the column layout)
//   Prover::build_constraint_commitment             prover/src/lib.rs:680-715
#pragma once

namespace wf {

// dst[k] = first ? src[k] : dst[k] + src[k] * factor      (k < n elements of E)
template <class F, int WE>
__global__ void __launch_bounds__(256) k_ext_scale_acc(typename F::T *__restrict__ dst, const typename F::T *__restrict__ src,
                                                       uint64_t n, Ext<F, WE> factor, int first) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const Ext<F, WE> v = ext_load<F, WE>(src + k * WE);
    if (first)
        ext_store<F, WE>(dst + k * WE, v);
    else
        ext_store<F, WE>(dst + k * WE, ext_add<F, WE>(ext_load<F, WE>(dst + k * WE), ext_mul<F, WE>(v, factor)));
}

template <class F, int WE>
static int comb_polys_dev(wf_ctx *ctx, hipStream_t st, const wf_params *p, const void *const *evals, size_t n_tables, uint32_t log_ce,
                          const void *final_coeff, void *d_polys) {
    typedef typename F::T T;
    typedef Ext<F, WE> E;
    const size_t ce = (size_t)1 << log_ce, bytes = ce * WE * sizeof(T);
    const uint64_t keep = ((uint64_t)p->n_cols) << p->log2_trace_len;  // coefficients that become columns
    int rc;
    if ((rc = ensure(ctx, ctx->io[0], bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[1], bytes))) return rc;
    u128 off;
    memcpy(&off, p->domain_offset, 16);
    TableSet *ser;
    if ((rc = series_tables<F>(ctx, log_ce, F::from_u128_canonical(off), (uint64_t)off, (uint64_t)(off >> 64), &ser))) return rc;
    E fc, pw;  // final_coeff and its running power
    for (int w = 0; w < WE; w++) {
        fc.c[w] = final_coeff ? ((const T *)final_coeff)[w] : F::zero();
        pw.c[w] = w == 0 ? F::one() : F::zero();
        if (!F::is_valid(fc.c[w])) return fail(WF_ERR_ARG, "final_coeff is not a valid field element");
    }
    for (size_t i = 0; i < n_tables; i++) {
        HIP_TRY(hipMemcpyAsync(ctx->io[0].p, evals[i], bytes, hipMemcpyHostToDevice, st));
        XformDesc<F> d;
        memset(&d, 0, sizeof(d));
        d.src = (const T *)ctx->io[0].p;
        d.dst = (T *)ctx->io[1].p;
        d.logN = log_ce;
        d.W = WE;
        d.batch = 1;
        d.inverse = true;
        d.scale_mode = SCALE_SERIES;
        d.out_series = ser;
        if ((rc = run_transform<F>(ctx, st, d))) return rc;
        prof_mark(ctx, st, "constraint.combine");
        hipLaunchKernelGGL((k_ext_scale_acc<F, WE>), dim3((uint32_t)((keep + 255) / 256)), dim3(256), 0, st, (T *)d_polys,
                           (const T *)ctx->io[1].p, keep, pw, i == 0 ? 1 : 0);
        HIP_TRY(hipGetLastError());
        prof_mark(ctx, st, "between_calls");
        pw = ext_mul<F, WE>(pw, fc);  // final_coeff.exp_vartime(i + 1)
    }
    return 0;
}

template <class F>
static int comb_polys_dispatch(wf_ctx *ctx, hipStream_t st, const wf_params *p, const void *const *evals, size_t n_tables,
                               uint32_t log_ce, const void *final_coeff, void *d_polys) {
    switch (p->ext_degree) {
        case 1: return comb_polys_dev<F, 1>(ctx, st, p, evals, n_tables, log_ce, final_coeff, d_polys);
        case 2: return comb_polys_dev<F, 2>(ctx, st, p, evals, n_tables, log_ce, final_coeff, d_polys);
        default:
            if constexpr (F::FIELD_ID == 1) return comb_polys_dev<F, 3>(ctx, st, p, evals, n_tables, log_ce, final_coeff, d_polys);
            return fail(WF_ERR_EXTENSION, "f128 has no cubic extension");
    }
}

}  // namespace wf

using namespace wf;

extern "C" {

int wf_constraint_commit_from_evaluations(wf_ctx *ctx, const wf_params *p, const void *const *combined_evaluations, size_t n_tables,
                                          size_t ce_domain_size, const void *final_coeff, void *const *polys_out, wf_commitment **out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!out) return fail(WF_ERR_ARG, "out is null");
    int rc = check_params(p, true);
    if (rc) return rc;
    if (!combined_evaluations || n_tables == 0) return fail(WF_ERR_ARG, "no evaluation tables");
    if (n_tables > 1 && !final_coeff) return fail(WF_ERR_ARG, "final_coeff is null");
    for (size_t i = 0; i < n_tables; i++)
        if (!combined_evaluations[i]) return fail(WF_ERR_ARG, "evaluation table %zu is null", i);
    const size_t R = (size_t)1 << p->log2_trace_len;
    // CompositionPoly::new: the size is a power of two larger than the trace length (composition_poly.rs:22-36); the
    // columns taken from it must exist (segment's chunks(trace_len).take(num_cols))
    if (ce_domain_size & (ce_domain_size - 1)) return fail(WF_ERR_TRACE_LENGTH, "size of composition polynomial must be a power of 2");
    if (ce_domain_size <= R) return fail(WF_ERR_TRACE_LENGTH, "trace length must be smaller than size of composition polynomial");
    if ((size_t)p->n_cols * R > ce_domain_size)
        return fail(WF_ERR_WIDTH, "%u columns of 2^%u coefficients do not fit a polynomial of %zu", p->n_cols, p->log2_trace_len, ce_domain_size);
    uint32_t log_ce = 0;
    while (((size_t)1 << log_ce) < ce_domain_size) log_ce++;
    const uint32_t adicity = p->field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY;
    if (log_ce > adicity) return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", log_ce);
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    wf_commitment *c = nullptr;
    bool dense = false;
    if ((rc = commitment_alloc(ctx, p, true, &c, &dense))) return rc;
    hipStream_t st = ctx->stream;
    rc = p->field == WF_FIELD_F64 ? comb_polys_dispatch<F64>(ctx, st, p, combined_evaluations, n_tables, log_ce, final_coeff, c->polys)
                                  : comb_polys_dispatch<F128>(ctx, st, p, combined_evaluations, n_tables, log_ce, final_coeff, c->polys);
    if (rc == 0)
        rc = p->field == WF_FIELD_F64 ? constraint_commit_dev<F64>(ctx, p, c->polys, c->lde, c->leaves, c->nodes, st, dense)
                                      : constraint_commit_dev<F128>(ctx, p, c->polys, c->lde, c->leaves, c->nodes, st, dense);
    if (rc == 0 && polys_out) rc = download_columns(ctx, polys_out, c->polys, p->n_cols, wf_column_bytes(p), st);
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(c->root, (char *)c->nodes + 32, 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(WF_ERR_HIP, "commitment failed: %s", hipGetErrorString(e));
    }
    if (rc) {
        (void)hipStreamSynchronize(st);  // queued copies read the caller's tables
        free_commitment(c);
        return rc;
    }
    *out = c;
    return 0;
}

}  // extern "C"
