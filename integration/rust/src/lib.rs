//! winter-hip: the two overridable methods of `trait Prover` on an MI355X, through libwf_lde.so.
//!
//! ```ignore
//! impl Prover for MyProver {
//!     type BaseField = math::fields::f64::BaseElement;          // or f128::BaseElement
//!     type HashFn = crypto::hashers::Blake3_256<Self::BaseField>;
//!     // get_pub_inputs / options / new_evaluator ... as before
//!
//!     fn build_trace_commitment<E>(&self, traces: Vec<&ColMatrix<E>>, domain: &StarkDomain<Self::BaseField>)
//!         -> (Vec<RowMatrix<E>>, MerkleTree<Self::HashFn>, Vec<ColMatrix<E>>)
//!     where E: FieldElement<BaseField = Self::BaseField>,
//!     {
//!         winter_hip::build_trace_commitment::<_, E, Self::HashFn>(&self.gpu, traces, domain)
//!     }
//!
//!     fn build_constraint_commitment<E>(&self, poly: &CompositionPoly<E>, domain: &StarkDomain<Self::BaseField>)
//!         -> ConstraintCommitment<E, Self::HashFn>
//!     where E: FieldElement<BaseField = Self::BaseField>,
//!     {
//!         winter_hip::build_constraint_commitment::<_, E, Self::HashFn>(&self.gpu, poly, domain)
//!     }
//! }
//! ```
//! replaces `prover/src/lib.rs:615-670` and `:680-715`.  The outputs are byte-identical to the CPU path's: f64 elements
//! cross the boundary as the Montgomery residues the reference keeps in memory (`math/src/field/f64/mod.rs:48-53`),
//! f128 as canonical `u128`, extension elements as consecutive base elements (`extensions/quadratic.rs:26-28`).
//!
//! Needs `RowMatrix::from_raw_parts` (winter-prover-rowmatrix.patch).  This crate has NOT been compiled: the
//! environment that produced libwf_lde.so has no Rust toolchain.  What is tested there is the C ABI itself
//! (tests/c/test_abi.c, the ctypes and C++ clients).

pub mod ffi;

use core::marker::PhantomData;
use std::ffi::CStr;
use std::os::raw::c_void;

use air::DeepCompositionCoefficients;
use crypto::{BatchMerkleProof, ElementHasher, Hasher, MerkleTree};
use math::{fields::f128, fields::f64, FieldElement, StarkField};
use prover::{ColMatrix, CompositionPoly, ConstraintCommitment, RowMatrix, StarkDomain};
use utils::{collections::Vec, uninit_vector, Deserializable, Serializable};

use ffi::*;

// FIELDS
// ================================================================================================

/// Base fields libwf_lde.so implements, with the id it knows them by.
pub trait WfField: StarkField {
    const WF_FIELD: u32;
}
impl WfField for f64::BaseElement {
    const WF_FIELD: u32 = WF_FIELD_F64;
}
impl WfField for f128::BaseElement {
    const WF_FIELD: u32 = WF_FIELD_F128;
}

/// Hashers the library implements: the digest is a byte array of `DIGEST_BYTES` bytes, written by the library as such
/// (`ByteDigest<N>`, `crypto/src/hash/mod.rs:84-85`): `Blake3_256` (32) and `Blake3_192` (24: the BLAKE3 output truncated,
/// 48-byte merges, `crypto/src/hash/blake/mod.rs:68-114`).  Host arrays of digests hold `DIGEST_BYTES` per entry.
pub trait WfHasher: ElementHasher + Hasher {
    const DIGEST_BYTES: u32;
}
impl<B: StarkField> WfHasher for crypto::hashers::Blake3_256<B> {
    const DIGEST_BYTES: u32 = 32;
}
impl<B: StarkField> WfHasher for crypto::hashers::Blake3_192<B> {
    const DIGEST_BYTES: u32 = 24;
}

// CONTEXT
// ================================================================================================

/// One per GPU (`wf_ctx`): twiddle tables, scratch and a stream.  Serves one call at a time; a second thread entering
/// gets `WF_ERR_BUSY`, so a prover shared between threads wraps it in a `Mutex` or keeps one context per thread.
pub struct WfContext {
    raw: *mut WfCtx,
}
unsafe impl Send for WfContext {}

impl WfContext {
    pub fn new(device: i32) -> Result<Self, String> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { wf_ctx_create(device, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn as_ptr(&self) -> *mut WfCtx {
        self.raw
    }
}
impl Drop for WfContext {
    fn drop(&mut self) {
        unsafe { wf_ctx_destroy(self.raw) }
    }
}

fn check(rc: i32) -> Result<(), String> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(wf_last_error()) }.to_string_lossy().into_owned();
    Err(format!("libwf_lde status {rc}: {msg}"))
}

fn params<B: WfField, E: FieldElement<BaseField = B>>(
    trace_len: usize, n_cols: usize, n_traces: usize, domain: &StarkDomain<B>, digest_bytes: u32,
) -> WfParams {
    let mut p = WfParams {
        field: B::WF_FIELD,
        ext_degree: E::EXTENSION_DEGREE as u32,
        log2_trace_len: trace_len.ilog2(),
        log2_blowup: domain.trace_to_lde_blowup().ilog2(),
        n_cols: n_cols as u32,
        n_traces: n_traces as u32,
        digest_bytes,
        reserved: 0,
        domain_offset: [0; 16],
    };
    // Serializable for a base element writes its canonical little-endian integer (f64/mod.rs:605-610, f128 likewise)
    let off = domain.offset().to_bytes();
    p.domain_offset[..off.len()].copy_from_slice(&off);
    p
}

// THE TWO METHODS, COPY-OUT FORM
// ================================================================================================

/// `Prover::build_trace_commitment` (prover/src/lib.rs:615-670): interpolate every column of every trace, evaluate over
/// the LDE domain into row-major matrices, hash the combined rows (STARKPack: one tree over all traces), build the tree.
pub fn build_trace_commitment<B, E, H>(
    ctx: &WfContext, traces: Vec<&ColMatrix<E>>, domain: &StarkDomain<B>,
) -> (Vec<RowMatrix<E>>, MerkleTree<H>, Vec<ColMatrix<E>>)
where
    B: WfField,
    E: FieldElement<BaseField = B>,
    H: WfHasher,
{
    assert!(!traces.is_empty(), "at least one trace is required");
    let (r, c, n) = (traces[0].num_rows(), traces[0].num_cols(), traces.len());
    assert!(traces.iter().all(|t| t.num_rows() == r && t.num_cols() == c), "packed traces must have one shape");
    let p = params::<B, E>(r, c, n, domain, H::DIGEST_BYTES);
    let row_width = unsafe { wf_row_width(&p) };
    let lde_rows = r * domain.trace_to_lde_blowup();

    // inputs: one pointer per column, [trace][col]; the elements are already in the library's representation
    let col_ptrs: Vec<*const c_void> = traces
        .iter()
        .flat_map(|t| (0..c).map(move |i| t.get_column(i).as_ptr() as *const c_void))
        .collect();
    // outputs, owned by Rust
    let mut polys: Vec<Vec<Vec<E>>> =
        (0..n).map(|_| (0..c).map(|_| unsafe { uninit_vector(r) }).collect()).collect();
    let mut ldes: Vec<Vec<B>> = (0..n).map(|_| unsafe { uninit_vector(lde_rows * row_width) }).collect();
    let mut leaves: Vec<H::Digest> = unsafe { uninit_vector(lde_rows) };
    let mut nodes: Vec<H::Digest> = unsafe { uninit_vector(lde_rows) };
    let poly_ptrs: Vec<*mut c_void> =
        polys.iter_mut().flat_map(|t| t.iter_mut().map(|v| v.as_mut_ptr() as *mut c_void)).collect();
    let lde_ptrs: Vec<*mut c_void> = ldes.iter_mut().map(|v| v.as_mut_ptr() as *mut c_void).collect();

    // (ByteDigest<32> is a [u8; 32] newtype; the reference itself views digest slices as bytes: ByteDigest::digests_as_bytes)
    let rc = unsafe {
        wf_trace_commit(
            ctx.raw, &p, col_ptrs.as_ptr(), poly_ptrs.as_ptr(), lde_ptrs.as_ptr(), leaves.as_mut_ptr() as *mut u8,
            nodes.as_mut_ptr() as *mut u8, core::ptr::null_mut(),
        )
    };
    check(rc).expect("failed to build trace commitment"); // the reference panics on the same preconditions

    let trace_ldes = ldes
        .into_iter()
        .map(|d| RowMatrix::from_raw_parts(d, row_width, c * E::EXTENSION_DEGREE))
        .collect();
    let tree = MerkleTree::from_raw_parts(nodes, leaves).expect("failed to construct trace Merkle tree");
    let trace_polys = polys.into_iter().map(ColMatrix::new).collect();
    (trace_ldes, tree, trace_polys)
}

/// `Prover::build_constraint_commitment` (prover/src/lib.rs:680-715): composition-polynomial columns (coefficient form)
/// -> `RowMatrix::evaluate_polys_over::<8>` -> `commit_to_rows`.
pub fn build_constraint_commitment<B, E, H>(
    ctx: &WfContext, composition_poly: &CompositionPoly<E>, domain: &StarkDomain<B>,
) -> ConstraintCommitment<E, H>
where
    B: WfField,
    E: FieldElement<BaseField = B>,
    H: WfHasher,
{
    let data = composition_poly.data(); // ColMatrix<E>: num_cols columns of trace_length coefficients
    let (r, c) = (data.num_rows(), data.num_cols());
    let p = params::<B, E>(r, c, 1, domain, H::DIGEST_BYTES);
    let row_width = unsafe { wf_row_width(&p) };
    let lde_rows = r * domain.trace_to_lde_blowup();

    let col_ptrs: Vec<*const c_void> = (0..c).map(|i| data.get_column(i).as_ptr() as *const c_void).collect();
    let mut lde: Vec<B> = unsafe { uninit_vector(lde_rows * row_width) };
    let mut leaves: Vec<H::Digest> = unsafe { uninit_vector(lde_rows) };
    let mut nodes: Vec<H::Digest> = unsafe { uninit_vector(lde_rows) };
    let rc = unsafe {
        wf_constraint_commit(
            ctx.raw, &p, col_ptrs.as_ptr(), lde.as_mut_ptr() as *mut c_void, leaves.as_mut_ptr() as *mut u8,
            nodes.as_mut_ptr() as *mut u8, core::ptr::null_mut(),
        )
    };
    check(rc).expect("failed to build constraint commitment");

    let evaluations = RowMatrix::from_raw_parts(lde, row_width, data.num_base_cols());
    let commitment = MerkleTree::from_raw_parts(nodes, leaves).expect("failed to construct constraint Merkle tree");
    ConstraintCommitment::new(evaluations, commitment)
}

// RESIDENT FORM
// ================================================================================================

/// A commitment that stays in HBM (`wf_commitment`): what `TraceCommitment` / `ConstraintCommitment` own in the
/// reference (prover/src/trace/commitment.rs:21-26, constraints/commitment.rs:21-24), minus the 0.5-16 GiB copy over
/// PCIe.  Only the queried rows and their Merkle proof ever leave the device.
pub struct ResidentCommitment<'a, E: FieldElement, H: WfHasher> {
    raw: *mut WfCommitment,
    n_traces: usize,
    n_cols: usize,
    _ctx: PhantomData<&'a WfContext>, // destroyed before its context
    _types: PhantomData<(E, H)>,
}

impl<'a, B, E, H> ResidentCommitment<'a, E, H>
where
    B: WfField,
    E: FieldElement<BaseField = B>,
    H: WfHasher,
{
    /// build_trace_commitment, outputs resident; returns the commitment and the trace polynomials (needed on the host
    /// by the constraint evaluator of the unmodified reference; skip with `want_polys = false` when they are not).
    pub fn commit_traces(
        ctx: &'a WfContext, traces: &[&ColMatrix<E>], domain: &StarkDomain<B>, want_polys: bool,
    ) -> (Self, Vec<ColMatrix<E>>) {
        let (r, c, n) = (traces[0].num_rows(), traces[0].num_cols(), traces.len());
        let p = params::<B, E>(r, c, n, domain, H::DIGEST_BYTES);
        let col_ptrs: Vec<*const c_void> = traces
            .iter()
            .flat_map(|t| (0..c).map(move |i| t.get_column(i).as_ptr() as *const c_void))
            .collect();
        let mut polys: Vec<Vec<Vec<E>>> = if want_polys {
            (0..n).map(|_| (0..c).map(|_| unsafe { uninit_vector(r) }).collect()).collect()
        } else {
            Vec::new()
        };
        let poly_ptrs: Vec<*mut c_void> =
            polys.iter_mut().flat_map(|t| t.iter_mut().map(|v| v.as_mut_ptr() as *mut c_void)).collect();
        let mut raw = core::ptr::null_mut();
        let rc = unsafe {
            wf_trace_commit_resident(
                ctx.raw, &p, col_ptrs.as_ptr(), if want_polys { poly_ptrs.as_ptr() } else { core::ptr::null() }, &mut raw,
            )
        };
        check(rc).expect("failed to build trace commitment");
        let me = Self { raw, n_traces: n, n_cols: c, _ctx: PhantomData, _types: PhantomData };
        (me, polys.into_iter().map(ColMatrix::new).collect())
    }

    /// The same for a STREAM of proofs (STARKPack proves batch after batch): returns as soon as the columns are on their way
    /// and the kernels are queued -- the next call's upload runs on the context's copy stream under this call's kernels.
    /// `wait` (or `root`) completes the handle.  The traces' columns are ordinary (pageable) `Vec`s here: the library has
    /// staged them when this returns, so `traces` may be dropped at once.
    pub fn commit_traces_async(ctx: &'a WfContext, traces: &[&ColMatrix<E>], domain: &StarkDomain<B>) -> Self {
        let (r, c, n) = (traces[0].num_rows(), traces[0].num_cols(), traces.len());
        let p = params::<B, E>(r, c, n, domain, H::DIGEST_BYTES);
        let col_ptrs: Vec<*const c_void> = traces
            .iter()
            .flat_map(|t| (0..c).map(move |i| t.get_column(i).as_ptr() as *const c_void))
            .collect();
        let mut raw = core::ptr::null_mut();
        check(unsafe { wf_trace_commit_resident_async(ctx.raw, &p, col_ptrs.as_ptr(), &mut raw) })
            .expect("failed to queue trace commitment");
        Self { raw, n_traces: n, n_cols: c, _ctx: PhantomData, _types: PhantomData }
    }

    /// Blocks until the kernels of an asynchronous commitment have finished (no-op otherwise).
    pub fn wait(&self) {
        check(unsafe { wf_commitment_wait(self.raw) }).expect("trace commitment failed on the device");
    }

    /// build_constraint_commitment, outputs resident.
    pub fn commit_composition_poly(ctx: &'a WfContext, poly: &CompositionPoly<E>, domain: &StarkDomain<B>) -> Self {
        let data = poly.data();
        let (r, c) = (data.num_rows(), data.num_cols());
        let p = params::<B, E>(r, c, 1, domain, H::DIGEST_BYTES);
        let col_ptrs: Vec<*const c_void> = (0..c).map(|i| data.get_column(i).as_ptr() as *const c_void).collect();
        let mut raw = core::ptr::null_mut();
        check(unsafe { wf_constraint_commit_resident(ctx.raw, &p, col_ptrs.as_ptr(), &mut raw) })
            .expect("failed to build constraint commitment");
        Self { raw, n_traces: 1, n_cols: c, _ctx: PhantomData, _types: PhantomData }
    }

    /// `MerkleTree::root` (crypto/src/merkle/mod.rs:167) -> `channel.commit_trace` / `commit_constraints`.
    pub fn root(&self) -> H::Digest {
        // root_out[32] is the digest zero-padded (Digest::as_bytes); H::Digest is ByteDigest<32> or ByteDigest<24>: read the
        // first DIGEST_BYTES through the type's own Deserializable impl (crypto/src/hash/mod.rs:127-131)
        let mut out = [0u8; 32];
        check(unsafe { wf_commitment_root(self.raw, out.as_mut_ptr()) }).expect("root");
        H::Digest::read_from_bytes(&out[..H::DIGEST_BYTES as usize]).expect("digest")
    }

    /// `TraceCommitment::query` / `ConstraintCommitment::query` (prover/src/trace/commitment.rs:87-111,
    /// constraints/commitment.rs:54-69): for every position the row of every trace (the `comb_states` that were hashed
    /// into the leaf: trace 0 || trace 1 || ..) and ONE batch proof -- one host round trip.
    pub fn query(&self, positions: &[usize]) -> (Vec<Vec<E>>, BatchMerkleProof<H>) {
        let pos: Vec<u64> = positions.iter().map(|&p| p as u64).collect();
        let (mut n_rows, mut row_elems, mut depth) = (0u64, 0u64, 0u32);
        check(unsafe { wf_commitment_info(self.raw, &mut n_rows, &mut row_elems, &mut depth) }).expect("info");
        let n = pos.len();
        // row_elems base elements per position = n_traces * n_cols elements of E
        let mut rows: Vec<E> = unsafe { uninit_vector(n * self.n_traces * self.n_cols) };
        let mut leaves: Vec<H::Digest> = unsafe { uninit_vector(n) };
        let cap = n * (depth as usize + 1);
        let mut flat: Vec<H::Digest> = unsafe { uninit_vector(cap) };
        let mut counts = vec![0u32; n.max(1)];
        let (mut n_vec, mut n_nodes, mut d) = (0usize, 0usize, 0u32);
        let rc = unsafe {
            wf_commitment_query(
                self.raw, pos.as_ptr(), n, rows.as_mut_ptr() as *mut c_void, leaves.as_mut_ptr() as *mut u8,
                flat.as_mut_ptr() as *mut u8, cap, counts.as_mut_ptr(), &mut n_vec, &mut n_nodes, &mut d,
            )
        };
        check(rc).expect("failed to query the commitment"); // TooFewLeafIndexes / duplicates / out of range, as the reference
        let mut nodes = Vec::with_capacity(n_vec);
        let mut k = 0;
        for &cnt in &counts[..n_vec] {
            nodes.push(flat[k..k + cnt as usize].to_vec());
            k += cnt as usize;
        }
        let per = self.n_traces * self.n_cols;
        let states = rows.chunks(per).map(|r| r.to_vec()).collect();
        (states, BatchMerkleProof { leaves, nodes, depth: d as u8 })
    }

    /// `TracePolyTable::get_ood_frame` half (prover/src/trace/poly_table.rs:67-70) / `CompositionPoly::evaluate_at`:
    /// every polynomial of the commitment at z; call twice (z, z * g) for a frame.
    pub fn evaluate_polys_at(&self, z: E) -> Vec<E> {
        let mut out: Vec<E> = unsafe { uninit_vector(self.n_traces * self.n_cols) };
        let rc = unsafe {
            wf_commitment_evaluate_polys_at(
                self.raw, &z as *const E as *const c_void, E::EXTENSION_DEGREE as u32, out.as_mut_ptr() as *mut c_void,
            )
        };
        check(rc).expect("failed to evaluate polynomials");
        out
    }
}

impl<'a, B, E, H> ResidentCommitment<'a, E, H>
where
    B: WfField,
    E: FieldElement<BaseField = B>,
    H: WfHasher,
{
    /// All of `ConstraintEvaluationTable::into_comb_poly` for every packed trace (division by the divisors, offset
    /// interpolation), STARKPack's `final_coeff` combination, `into_poly` and `build_constraint_commitment`
    /// (prover/src/lib.rs:435-472) in one call, resident.  `tables[i]` = the evaluation columns of packed trace i with their
    /// divisors ((numerator degree a, numerator constant b, exemption points): `ConstraintDivisor` has one numerator term).
    pub fn commit_evaluation_tables(
        ctx: &'a WfContext, tables: &[Vec<(&[E], (u64, B, Vec<B>))>], trace_length: usize, num_cols: usize, final_coeff: E,
        domain: &StarkDomain<B>,
    ) -> Self {
        let ce = tables[0][0].0.len();
        let p = params::<B, E>(trace_length, num_cols, 1, domain, H::DIGEST_BYTES);
        let mut col_ptrs: Vec<Vec<*const c_void>> = Vec::new();
        let mut divisors: Vec<Vec<WfDivisor>> = Vec::new();
        for t in tables {
            col_ptrs.push(t.iter().map(|(c, _)| c.as_ptr() as *const c_void).collect());
            divisors.push(
                t.iter()
                    .map(|(_, (a, b, ex))| {
                        let mut constant = [0u8; 16];
                        let raw = unsafe { core::slice::from_raw_parts(b as *const B as *const u8, core::mem::size_of::<B>()) };
                        constant[..raw.len()].copy_from_slice(raw);
                        WfDivisor {
                            numerator_degree: *a,
                            numerator_constant: constant,
                            exemptions: ex.as_ptr() as *const c_void,
                            n_exemptions: ex.len() as u32,
                        }
                    })
                    .collect(),
            );
        }
        let c_tables: Vec<WfEvaluationTable> = col_ptrs
            .iter()
            .zip(divisors.iter())
            .map(|(c, d)| WfEvaluationTable { columns: c.as_ptr(), divisors: d.as_ptr(), n_columns: c.len() as u32 })
            .collect();
        let mut raw = core::ptr::null_mut();
        let rc = unsafe {
            wf_constraint_commit_from_tables(
                ctx.raw, &p, c_tables.as_ptr(), c_tables.len(), ce, &final_coeff as *const E as *const c_void, core::ptr::null(),
                &mut raw,
            )
        };
        check(rc).expect("failed to build constraint commitment");
        Self { raw, n_traces: 1, n_cols: num_cols, _ctx: PhantomData, _types: PhantomData }
    }

    /// The rows of the constraint evaluation domain (every `stride`-th LDE row, `stride` = lde blowup / ce blowup) of one
    /// trace's extended matrix, `row_width` base elements each, for the AIR's evaluator on the host
    /// (`TraceLde::read_main_trace_frame_into`, prover/src/trace/trace_lde.rs:78-98).
    pub fn read_ce_rows(&self, trace: usize, stride: usize) -> (Vec<B>, usize) {
        let (mut n_rows, mut row_elems, mut depth, mut width) = (0u64, 0u64, 0u32, 0u64);
        check(unsafe { wf_commitment_info(self.raw, &mut n_rows, &mut row_elems, &mut depth) }).expect("info");
        check(unsafe { wf_commitment_read_lde(self.raw, trace as u32, 0, 0, core::ptr::null_mut(), &mut width) }).expect("width");
        let n = n_rows as usize / stride;
        let mut rows: Vec<B> = unsafe { uninit_vector(n * width as usize) };
        let rc = unsafe {
            wf_commitment_read_lde_strided(
                self.raw, trace as u32, 0, n as u64, stride as u64, rows.as_mut_ptr() as *mut c_void, &mut width,
            )
        };
        check(rc).expect("failed to read the extended trace");
        (rows, width as usize)
    }
}

/// `DeepCompositionPoly::new(z, cc)` + `add_trace_polys` + `add_composition_poly` + `evaluate(&domain)` +
/// `fri_prover.build_layers`'s first step (prover/src/lib.rs:508-560): the composition over the polynomials the resident
/// commitments hold, its LDE and the FRI prover's first layer, all in HBM.  `cc.traces` flattened handle by handle, trace by
/// trace, column by column.  The out-of-domain frames are not arguments: the composer's result does not depend on them
/// (`wf_lde.h`, wf_deep_compose).
pub fn deep_compose_into_fri<B, E, H>(
    ctx: &WfContext, traces: &[&ResidentCommitment<E, H>], constraints: &ResidentCommitment<E, H>, z: E,
    cc: &DeepCompositionCoefficients<E>, fri: *mut WfFriProver, lde_blowup: usize,
) where
    B: WfField,
    E: FieldElement<BaseField = B>,
    H: WfHasher,
{
    let handles: Vec<*const WfCommitment> = traces.iter().map(|t| t.raw as *const WfCommitment).collect();
    let flat: Vec<E> = cc.traces.iter().flatten().copied().collect();
    let rc = unsafe {
        wf_deep_compose(
            ctx.raw, handles.as_ptr(), handles.len(), constraints.raw, &z as *const E as *const c_void, E::EXTENSION_DEGREE as u32,
            flat.as_ptr() as *const c_void, cc.constraints.as_ptr() as *const c_void, core::ptr::null_mut(), fri, lde_blowup,
        )
    };
    check(rc).expect("failed to build the DEEP composition polynomial");
}

impl<'a, E: FieldElement, H: WfHasher> Drop for ResidentCommitment<'a, E, H> {
    fn drop(&mut self) {
        unsafe { wf_commitment_destroy(self.raw) }
    }
}
