// Constraint side of the path from the constraint evaluation TABLE (or its already combined column) to the resident
// constraint commitment without the composition polynomial visiting the host (SURVEY.md §8 rows a19-a21 chained):
//   ConstraintEvaluationTable::into_comb_poly      /root/reference/prover/src/constraints/evaluation_table.rs:166-186
//       (acc_column :335-391 + get_inv_evaluation :393-426: every column divided by its divisor and summed -- generic in
//        the divisor's numerator (x^a - b) and exemption points, nothing AIR-specific left in it -- then
//        fft::interpolate_poly_with_offset over the constraint evaluation domain)
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

// ---- ConstraintEvaluationTable::into_comb_poly in front of the interpolation (evaluation_table.rs:166-176, 335-426):
// every column of the table divided by its divisor (x^a - b) / prod_k (x - e_k) over the constraint evaluation domain
// x_i = offset * g^i and summed.  z_j = 1 / (x_j^a - b) takes ce / a distinct values (get_inv_evaluation).
constexpr uint32_t MAX_EXEMPTIONS = 8;

template <class F>
__device__ __forceinline__ typename F::T dev_inv(typename F::T x) {  // x^(p - 2); 0 -> 0 like math::batch_inversion
    if constexpr (F::FIELD_ID == 1)
        return f_pow<F>(x, (u128)(F64::P - 2));
    else
        return f_pow<F>(x, F128::P() - 2);
}

// z[j] = 1 / (offset^a * g^((a j) mod ce) - b), j < nz = ce / a   (get_ce_x_power_at, domain.rs:134-142)
template <class F>
__global__ void __launch_bounds__(256) k_divisor_inverses(typename F::T *__restrict__ z, uint64_t nz, uint64_t a, uint64_t ce_mask,
                                                          typename F::T offset_exp, typename F::T b, Pow2L<F> g) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nz) return;
    z[j] = dev_inv<F>(F::sub(F::mul(g.get((j * a) & ce_mask), offset_exp), b));
}

template <class F>
struct AccColumnArgs {
    typedef typename F::T T;
    T *acc;          // [ce] elements of E
    const T *col;    // [ce] elements of E
    const T *z;      // [nz]
    uint64_t ce, nz_mask;
    uint32_t n_ex, first;  // first: acc is written, not added to (E::zeroed_vector + the first column)
    T ex[MAX_EXEMPTIONS];
    T offset;
    Pow2L<F> g;      // powers of the ce domain's generator
};

// acc[i] (+)= col[i].mul_base(z[i mod nz] * prod_k (x_i - ex_k))      (acc_column, evaluation_table.rs:335-391)
template <class F, int WE>
__global__ void __launch_bounds__(256) k_acc_column(AccColumnArgs<F> a) {
    typedef typename F::T T;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.ce) return;
    T f = a.z[i & a.nz_mask];
    if (a.n_ex) {
        const T x = F::mul(a.g.get(i), a.offset);  // get_ce_x_at
        T e = F::sub(x, a.ex[0]);                  // evaluate_exemptions_at (divisor.rs:148-152)
        for (uint32_t k = 1; k < a.n_ex; k++) e = F::mul(e, F::sub(x, a.ex[k]));
        f = F::mul(f, e);
    }
    Ext<F, WE> v = ext_load<F, WE>(a.col + i * WE);
#pragma unroll
    for (int w = 0; w < WE; w++) v.c[w] = F::mul(v.c[w], f);
    if (!a.first) v = ext_add<F, WE>(v, ext_load<F, WE>(a.acc + i * WE));
    ext_store<F, WE>(a.acc + i * WE, v);
}

// the table's columns -> its combined column in ctx->io[2] (device)
template <class F, int WE>
static int combine_table_dev(wf_ctx *ctx, hipStream_t st, const wf_params *p, const wf_evaluation_table *tab, uint32_t log_ce,
                             std::vector<std::vector<typename F::T>> &keep_alive) {
    typedef typename F::T T;
    const uint64_t ce = (uint64_t)1 << log_ce;
    const size_t bytes = ce * WE * sizeof(T);
    int rc;
    if ((rc = ensure(ctx, ctx->io[0], bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[2], bytes))) return rc;
    TableSet *tw;
    if ((rc = root_tables<F>(ctx, log_ce, false, &tw))) return rc;
    u128 off;
    memcpy(&off, p->domain_offset, 16);
    const T offset = F::from_u128_canonical(off);
    for (uint32_t j = 0; j < tab->n_columns; j++) {
        const wf_divisor &d = tab->divisors[j];
        const uint64_t a = d.numerator_degree;
        T b;
        memcpy(&b, d.numerator_constant, sizeof(T));
        const uint64_t nz = ce / a;
        const T offset_exp = f_pow<F>(offset, (u128)a);
        // the inverses: a handful for a transition divisor (ce / trace length values): on the host; as many as the domain
        // has points for an assertion at a single step (a = 1): on the device
        if ((rc = ensure(ctx, ctx->io[3], nz * sizeof(T)))) return rc;
        if (nz <= 1024) {
            keep_alive.emplace_back(nz);
            std::vector<T> &z = keep_alive.back();
            const T g = f_root_of_unity<F>(log_ce);
            for (uint64_t q = 0; q < nz; q++) {
                const T v = F::sub(F::mul(f_pow<F>(g, (u128)((q * a) & (ce - 1))), offset_exp), b);
                z[q] = f_inv<F>(v);  // 0 -> 0
            }
            HIP_TRY(hipMemcpyAsync(ctx->io[3].p, z.data(), nz * sizeof(T), hipMemcpyHostToDevice, st));
        } else {
            hipLaunchKernelGGL(k_divisor_inverses<F>, dim3((uint32_t)((nz + 255) / 256)), dim3(256), 0, st, (T *)ctx->io[3].p, nz, a,
                               ce - 1, offset_exp, b, as_pow2l<F>(*tw));
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(ctx->io[0].p, tab->columns[j], bytes, hipMemcpyHostToDevice, st));
        AccColumnArgs<F> ka;
        memset(&ka, 0, sizeof(ka));
        ka.acc = (T *)ctx->io[2].p;
        ka.col = (const T *)ctx->io[0].p;
        ka.z = (const T *)ctx->io[3].p;
        ka.ce = ce;
        ka.nz_mask = nz - 1;
        ka.n_ex = d.n_exemptions;
        ka.first = j == 0;
        for (uint32_t k = 0; k < d.n_exemptions; k++) memcpy(&ka.ex[k], (const char *)d.exemptions + (size_t)k * sizeof(T), sizeof(T));
        ka.offset = offset;
        ka.g = as_pow2l<F>(*tw);
        prof_mark(ctx, st, "constraint.acc_column");
        hipLaunchKernelGGL((k_acc_column<F, WE>), dim3((uint32_t)((ce + 255) / 256)), dim3(256), 0, st, ka);
        HIP_TRY(hipGetLastError());
        prof_mark(ctx, st, "between_calls");
    }
    return 0;
}

template <class F, int WE>
static int comb_polys_dev(wf_ctx *ctx, hipStream_t st, const wf_params *p, const void *const *evals, const wf_evaluation_table *tables,
                          size_t n_tables, uint32_t log_ce, const void *final_coeff, void *d_polys,
                          std::vector<std::vector<typename F::T>> &keep_alive) {
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
        const T *d_combined = (const T *)ctx->io[0].p;
        if (tables) {  // the table's columns divided by their divisors and summed, on the device
            if ((rc = combine_table_dev<F, WE>(ctx, st, p, &tables[i], log_ce, keep_alive))) return rc;
            d_combined = (const T *)ctx->io[2].p;
        } else {
            HIP_TRY(hipMemcpyAsync(ctx->io[0].p, evals[i], bytes, hipMemcpyHostToDevice, st));
        }
        XformDesc<F> d;
        memset(&d, 0, sizeof(d));
        d.src = d_combined;
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
static int comb_polys_dispatch(wf_ctx *ctx, hipStream_t st, const wf_params *p, const void *const *evals,
                               const wf_evaluation_table *tables, size_t n_tables, uint32_t log_ce, const void *final_coeff, void *d_polys) {
    std::vector<std::vector<typename F::T>> keep_alive;  // host tables read by queued copies
    int rc;
    switch (p->ext_degree) {
        case 1: rc = comb_polys_dev<F, 1>(ctx, st, p, evals, tables, n_tables, log_ce, final_coeff, d_polys, keep_alive); break;
        case 2: rc = comb_polys_dev<F, 2>(ctx, st, p, evals, tables, n_tables, log_ce, final_coeff, d_polys, keep_alive); break;
        default:
            if constexpr (F::FIELD_ID == 1)
                rc = comb_polys_dev<F, 3>(ctx, st, p, evals, tables, n_tables, log_ce, final_coeff, d_polys, keep_alive);
            else
                rc = fail(WF_ERR_EXTENSION, "f128 has no cubic extension");
    }
    if (!keep_alive.empty()) (void)hipStreamSynchronize(st);
    return rc;
}

}  // namespace wf

using namespace wf;

static int constraint_commit_from_impl(wf_ctx *ctx, const wf_params *p, const void *const *combined_evaluations,
                                       const wf_evaluation_table *tables, size_t n_tables, size_t ce_domain_size, const void *final_coeff,
                                       void *const *polys_out, wf_commitment **out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!out) return fail(WF_ERR_ARG, "out is null");
    int rc = check_params(p, true);
    if (rc) return rc;
    if ((!combined_evaluations && !tables) || n_tables == 0) return fail(WF_ERR_ARG, "no evaluation tables");
    if (n_tables > 1 && !final_coeff) return fail(WF_ERR_ARG, "final_coeff is null");
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
    const size_t eb = wf_elem_bytes(p->field);
    for (size_t i = 0; i < n_tables; i++) {
        if (!tables) {
            if (!combined_evaluations[i]) return fail(WF_ERR_ARG, "evaluation table %zu is null", i);
            continue;
        }
        const wf_evaluation_table &t = tables[i];
        if (t.n_columns == 0 || !t.columns || !t.divisors) return fail(WF_ERR_ARG, "evaluation table %zu is empty", i);
        for (uint32_t j = 0; j < t.n_columns; j++) {
            const wf_divisor &d = t.divisors[j];
            if (!t.columns[j]) return fail(WF_ERR_ARG, "table %zu: column %u is null", i, j);
            // numerator degree: a power of two dividing the domain (trace length / assertion stride, divisor.rs:52-128)
            if (d.numerator_degree == 0 || (d.numerator_degree & (d.numerator_degree - 1)) || d.numerator_degree > ce_domain_size)
                return fail(WF_ERR_ARG, "table %zu: divisor %u has numerator degree %llu", i, j, (unsigned long long)d.numerator_degree);
            if (d.n_exemptions > MAX_EXEMPTIONS) return fail(WF_ERR_ARG, "table %zu: divisor %u has %u exemption points (at most %u)", i, j, d.n_exemptions, MAX_EXEMPTIONS);
            if (d.n_exemptions && !d.exemptions) return fail(WF_ERR_ARG, "table %zu: divisor %u: exemptions is null", i, j);
            bool ok = p->field == WF_FIELD_F64 ? F64::is_valid(*(const uint64_t *)d.numerator_constant) : true;
            if (p->field == WF_FIELD_F128) {
                U128 v;
                memcpy(&v, d.numerator_constant, 16);
                ok = F128::is_valid(v);
            }
            for (uint32_t k = 0; ok && k < d.n_exemptions; k++) {
                if (p->field == WF_FIELD_F64) {
                    ok = F64::is_valid(((const uint64_t *)d.exemptions)[k]);
                } else {
                    U128 v;
                    memcpy(&v, (const char *)d.exemptions + (size_t)k * eb, 16);
                    ok = F128::is_valid(v);
                }
            }
            if (!ok) return fail(WF_ERR_ARG, "table %zu: divisor %u holds an invalid field element", i, j);
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    wf_commitment *c = nullptr;
    bool dense = false;
    if ((rc = commitment_alloc(ctx, p, true, &c, &dense))) return rc;
    hipStream_t st = ctx->stream;
    rc = p->field == WF_FIELD_F64
             ? comb_polys_dispatch<F64>(ctx, st, p, combined_evaluations, tables, n_tables, log_ce, final_coeff, c->polys)
             : comb_polys_dispatch<F128>(ctx, st, p, combined_evaluations, tables, n_tables, log_ce, final_coeff, c->polys);
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

extern "C" {

int wf_constraint_commit_from_evaluations(wf_ctx *ctx, const wf_params *p, const void *const *combined_evaluations, size_t n_tables,
                                          size_t ce_domain_size, const void *final_coeff, void *const *polys_out, wf_commitment **out) {
    if (!combined_evaluations) return fail(WF_ERR_ARG, "no evaluation tables");
    return constraint_commit_from_impl(ctx, p, combined_evaluations, nullptr, n_tables, ce_domain_size, final_coeff, polys_out, out);
}

int wf_constraint_commit_from_tables(wf_ctx *ctx, const wf_params *p, const wf_evaluation_table *tables, size_t n_tables,
                                     size_t ce_domain_size, const void *final_coeff, void *const *polys_out, wf_commitment **out) {
    if (!tables) return fail(WF_ERR_ARG, "no evaluation tables");
    return constraint_commit_from_impl(ctx, p, nullptr, tables, n_tables, ce_domain_size, final_coeff, polys_out, out);
}

}  // extern "C"
