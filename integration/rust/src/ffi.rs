//! Raw declarations of the C ABI in include/wf_lde.h (the subset the binding uses; names and argument order are the
//! header's).  Status codes: 0 = success, negative = `wf_status`; `wf_last_error()` describes the calling thread's
//! last failure.

use std::os::raw::{c_char, c_int, c_void};

pub const WF_FIELD_F64: u32 = 1;
pub const WF_FIELD_F128: u32 = 2;
pub const WF_COMM_ID_BYTES: usize = 128;

/// `wf_params` (include/wf_lde.h): what StarkDomain + ProofOptions carry into the two Prover methods.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct WfParams {
    pub field: u32,
    pub ext_degree: u32,
    pub log2_trace_len: u32,
    pub log2_blowup: u32,
    pub n_cols: u32,
    pub n_traces: u32,
    pub digest_bytes: u32,
    pub reserved: u32,
    pub domain_offset: [u8; 16],
}

#[repr(C)]
pub struct WfCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct WfCommitment {
    _private: [u8; 0],
}
/// `wf_divisor`: (x^a - b) / prod_k (x - e_k) -- `ConstraintDivisor` with its single numerator term.
#[repr(C)]
pub struct WfDivisor {
    pub numerator_degree: u64,
    pub numerator_constant: [u8; 16],
    pub exemptions: *const c_void,
    pub n_exemptions: u32,
}
/// `wf_evaluation_table`: the columns of one `ConstraintEvaluationTable` with their divisors.
#[repr(C)]
pub struct WfEvaluationTable {
    pub columns: *const *const c_void,
    pub divisors: *const WfDivisor,
    pub n_columns: u32,
}
/// `wf_query` of `wf_lde.h`: one commitment's share of a batched query.
#[repr(C)]
pub struct WfQuery {
    pub commitment: *const WfCommitment,
    pub positions: *const u64,
    pub n: usize,
    pub rows_out: *mut c_void,
    pub leaves_out: *mut u8,
    pub nodes_out: *mut u8,
    pub nodes_capacity: usize,
    pub node_counts: *mut u32,
    pub n_vectors: usize,
    pub n_nodes: usize,
    pub depth: u32,
}

#[repr(C)]
pub struct WfFriProver {
    _private: [u8; 0],
}
#[repr(C)]
pub struct WfComm {
    _private: [u8; 0],
}

extern "C" {
    pub fn wf_last_error() -> *const c_char;
    pub fn wf_device_count() -> c_int;
    pub fn wf_ctx_create(device: c_int, out: *mut *mut WfCtx) -> c_int;
    pub fn wf_ctx_destroy(ctx: *mut WfCtx);
    pub fn wf_ctx_synchronize(ctx: *mut WfCtx) -> c_int;
    pub fn wf_row_width(p: *const WfParams) -> usize;

    // host-buffer form
    pub fn wf_trace_commit(
        ctx: *mut WfCtx, p: *const WfParams, trace_cols: *const *const c_void, polys_out: *const *mut c_void,
        lde_out: *const *mut c_void, leaves_out: *mut u8, nodes_out: *mut u8, root_out: *mut u8,
    ) -> c_int;
    pub fn wf_constraint_commit(
        ctx: *mut WfCtx, p: *const WfParams, poly_cols: *const *const c_void, lde_out: *mut c_void,
        leaves_out: *mut u8, nodes_out: *mut u8, root_out: *mut u8,
    ) -> c_int;

    // resident form
    pub fn wf_trace_commit_resident(
        ctx: *mut WfCtx, p: *const WfParams, trace_cols: *const *const c_void, polys_out: *const *mut c_void,
        out: *mut *mut WfCommitment,
    ) -> c_int;
    pub fn wf_constraint_commit_resident(
        ctx: *mut WfCtx, p: *const WfParams, poly_cols: *const *const c_void, out: *mut *mut WfCommitment,
    ) -> c_int;
    pub fn wf_commitment_destroy(c: *mut WfCommitment);
    pub fn wf_commitment_root(c: *const WfCommitment, root_out: *mut u8) -> c_int;
    pub fn wf_commitment_info(c: *const WfCommitment, n_rows: *mut u64, row_elems: *mut u64, depth: *mut u32) -> c_int;
    pub fn wf_commitment_query(
        c: *const WfCommitment, positions: *const u64, n: usize, rows_out: *mut c_void, leaves_out: *mut u8,
        nodes_out: *mut u8, nodes_capacity: usize, node_counts: *mut u32, n_vectors: *mut usize, n_nodes: *mut usize,
        depth_out: *mut u32,
    ) -> c_int;
    pub fn wf_commitment_read_lde(
        c: *const WfCommitment, trace: u32, row_begin: u64, n_rows: u64, rows_out: *mut c_void, row_width_out: *mut u64,
    ) -> c_int;
    pub fn wf_commitment_evaluate_polys_at_points(
        c: *const WfCommitment, points: *const c_void, n_points: u32, z_ext_degree: u32, out: *mut c_void,
    ) -> c_int;
    pub fn wf_commitment_query_many(queries: *mut WfQuery, n_queries: usize) -> c_int;
    pub fn wf_constraint_commit_from_evaluations(
        ctx: *mut WfCtx, p: *const WfParams, combined_evaluations: *const *const c_void, n_tables: usize, ce_domain_size: usize,
        final_coeff: *const c_void, polys_out: *const *mut c_void, out: *mut *mut WfCommitment,
    ) -> c_int;
    pub fn wf_constraint_commit_from_tables(
        ctx: *mut WfCtx, p: *const WfParams, tables: *const WfEvaluationTable, n_tables: usize, ce_domain_size: usize,
        final_coeff: *const c_void, polys_out: *const *mut c_void, out: *mut *mut WfCommitment,
    ) -> c_int;
    pub fn wf_deep_compose(
        ctx: *mut WfCtx, trace_commitments: *const *const WfCommitment, n_trace_commitments: usize,
        constraint_commitment: *const WfCommitment, z: *const c_void, ext_degree: u32, trace_coeffs: *const c_void,
        constraint_coeffs: *const c_void, poly_out: *mut c_void, fri: *mut WfFriProver, lde_blowup: usize,
    ) -> c_int;
    pub fn wf_commitment_read_lde_strided(
        c: *const WfCommitment, trace: u32, row_begin: u64, n_rows: u64, row_stride: u64, rows_out: *mut c_void,
        row_width_out: *mut u64,
    ) -> c_int;
    pub fn wf_commitment_evaluate_polys_at(
        c: *const WfCommitment, z: *const c_void, z_ext_degree: u32, out: *mut c_void,
    ) -> c_int;

    // FRI commit phase, resident
    pub fn wf_fri_prover_create(
        ctx: *mut WfCtx, field: u32, ext_degree: u32, folding: u32, blowup: u32, remainder_max_degree: u32,
        domain_offset: *const u8, out: *mut *mut WfFriProver,
    ) -> c_int;
    pub fn wf_fri_prover_destroy(pr: *mut WfFriProver);
    pub fn wf_fri_prover_begin(pr: *mut WfFriProver, evals: *const c_void, n: usize) -> c_int;
    pub fn wf_fri_prover_commit_layer(pr: *mut WfFriProver, root_out: *mut u8) -> c_int;
    pub fn wf_fri_prover_fold(pr: *mut WfFriProver, alpha: *const c_void) -> c_int;
    pub fn wf_fri_prover_set_remainder(
        pr: *mut WfFriProver, remainder_out: *mut c_void, capacity: usize, len_out: *mut usize, commitment_out: *mut u8,
    ) -> c_int;
    pub fn wf_fri_prover_layer(pr: *const WfFriProver, i: usize, out: *mut *const WfCommitment) -> c_int;
    pub fn wf_fri_prover_reset(pr: *mut WfFriProver) -> c_int;

    // several GPUs: one process (or thread) per device
    pub fn wf_comm_unique_id(id_out: *mut u8) -> c_int;
    pub fn wf_comm_create(ctx: *mut WfCtx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut WfComm) -> c_int;
    pub fn wf_comm_destroy(comm: *mut WfComm);
    pub fn wf_comm_all_gather_roots(
        comm: *mut WfComm, d_roots: *const c_void, n_roots: usize, d_all: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn wf_ctx_set_digest_bytes(ctx: *mut WfCtx, digest_bytes: u32) -> c_int;
    /// A stream of proofs: returns once the columns are on their way; `wf_commitment_wait` (or `wf_commitment_root`) completes it.
    pub fn wf_trace_commit_resident_async(
        ctx: *mut WfCtx, p: *const WfParams, trace_cols: *const *const c_void, out: *mut *mut WfCommitment,
    ) -> c_int;
    pub fn wf_commitment_wait(c: *mut WfCommitment) -> c_int;
    pub fn wf_comm_barrier(comm: *mut WfComm) -> c_int;
    /// One f64 of every rank on every rank (`all_out`: world values, rank-major): per-rank step times of a benchmark record.
    pub fn wf_comm_gather_f64(comm: *mut WfComm, value: f64, all_out: *mut f64) -> c_int;
    /// What the transport reports about the communicator: transport (0 = RCCL, 1 = caller-supplied) and, for RCCL,
    /// ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the ncclComm_t in use.  Any output may be null.
    pub fn wf_comm_info(
        comm: *const WfComm, transport: *mut c_int, nccl_count: *mut c_int, nccl_user_rank: *mut c_int, nccl_device: *mut c_int,
    ) -> c_int;
    /// Blocking wait on `stream` under the communicator's watchdog (WF_COMM_TIMEOUT_S): WF_ERR_COMM instead of a hang.
    pub fn wf_comm_stream_wait(comm: *mut WfComm, stream: *mut c_void) -> c_int;
    /// The file the RCCL symbols were resolved from (NUL-terminated, static storage).
    pub fn wf_comm_rccl_path() -> *const std::os::raw::c_char;
    pub fn wf_trace_commit_sharded_dev(
        comm: *mut WfComm, p: *const WfParams, d_trace: *const c_void, d_polys: *mut c_void, d_lde_shard: *mut c_void,
        d_leaves: *mut c_void, d_nodes: *mut c_void, d_top: *mut c_void, stream: *mut c_void,
    ) -> c_int;
}
