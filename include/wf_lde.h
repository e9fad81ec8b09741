/*
 * wf_lde.h -- C ABI of libwf_lde.so: MI355X (gfx950) implementation of the winter-prover hot path
 *   trace / constraint low-degree extension (radix-2 NTT over the f64 and f128 base fields, coset evaluation)
 *   + BLAKE3-256 Merkle commitment of the extended rows, including STARKPack's combined-row commitment.
 *
 * The reference (Rust) has no FFI layer; the seam these entry points replace is the pair of provided methods
 *   Prover::build_trace_commitment       /root/reference/prover/src/lib.rs:615-670
 *   Prover::build_constraint_commitment  /root/reference/prover/src/lib.rs:680-715
 * plus the math::fft / crypto calls they are made of.  INTEGRATION.md shows the Rust `impl Prover` override that
 * binds them.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Element memory representation (identical to the reference's in-memory types):
 *   WF_FIELD_F64  : uint64_t Montgomery residue x*2^64 mod p, p = 2^64-2^32+1   (math/src/field/f64/mod.rs:48-53)
 *   WF_FIELD_F128 : 16-byte little-endian canonical integer < p = 2^128-45*2^40+1 (math/src/field/f128/mod.rs:35)
 *   extension element (ext_degree 2 or 3): ext_degree consecutive base elements  (extensions/quadratic.rs:26-28)
 * Column  = trace_len extension elements, contiguous (ColMatrix column, prover/src/matrix/col_matrix.rs:31).
 * RowMatrix data = (trace_len*blowup) rows x row_width base elements, row_width = 8*ceil(n_cols*ext_degree/8),
 *   unused lanes zero (prover/src/matrix/row_matrix.rs:31-39,118-133; segments.rs:65-72).
 * Digest = 32 bytes (crypto/src/hash/mod.rs:84-85).  Merkle `nodes` = n_leaves digests, nodes[0] = zero digest,
 *   nodes[1] = root, nodes[i] = BLAKE3(nodes[2i] || nodes[2i+1]) (crypto/src/merkle/mod.rs:87-90,350-374).
 *
 * All functions return 0 on success or a negative wf_status; wf_last_error() describes the last failure of the
 * calling thread.  Nothing here aborts or falls back to a CPU path: without a usable HIP device every compute
 * entry point fails with WF_ERR_HIP.
 */
#ifndef WF_LDE_H
#define WF_LDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wf_ctx wf_ctx;

enum wf_field { WF_FIELD_F64 = 1, WF_FIELD_F128 = 2 };

enum wf_status {
    WF_OK = 0,
    WF_ERR_FIELD = -10,        /* unknown field id */
    WF_ERR_EXTENSION = -11,    /* ext_degree not in {1,2,3}, or 3 over f128 (f128/mod.rs:296-314) */
    WF_ERR_TRACE_LENGTH = -12, /* trace length < 8 or not a power of two (air/src/air/trace_info.rs:35) */
    WF_ERR_BLOWUP = -13,       /* blowup not a power of two in [2,128] (air/src/options.rs:19-20) */
    WF_ERR_DOMAIN = -14,       /* log2(trace_len*blowup) exceeds the field's two-adicity (fft/mod.rs:196-200) */
    WF_ERR_WIDTH = -15,        /* n_cols not in [1,255] (air/src/air/trace_info.rs:37) */
    WF_ERR_TRACES = -16,       /* n_traces == 0 */
    WF_ERR_OFFSET = -17,       /* domain offset == 0 or >= p (fft/mod.rs:201) */
    WF_ERR_LEAVES = -18,       /* fewer than two leaves / not a power of two (merkle/mod.rs:118-123) */
    WF_ERR_ARG = -19,          /* null pointer or other malformed argument */
    WF_ERR_BUSY = -20,         /* the context is inside a call of another thread (one call at a time per wf_ctx) */
    WF_ERR_HIP = -30,          /* HIP runtime failure (no device, out of memory, launch failure) */
    WF_ERR_DIGEST = -31,       /* digest_bytes is neither 32 (Blake3_256) nor 24 (Blake3_192) */
    WF_ERR_COMM = -32          /* RCCL (or caller-supplied transport) failure, librccl.so.1 not loadable */
};

/* Parameters of one commitment.  Mirrors what StarkDomain + ProofOptions carry into the two Prover methods
 * (prover/src/domain.rs:13-31; air/src/options.rs:199-201). */
typedef struct wf_params {
    uint32_t field;          /* enum wf_field */
    uint32_t ext_degree;     /* E::EXTENSION_DEGREE: 1, 2 or 3 */
    uint32_t log2_trace_len; /* log2 of the trace length R (polynomial size) */
    uint32_t log2_blowup;    /* log2 of StarkDomain::trace_to_lde_blowup() */
    uint32_t n_cols;         /* columns of E per trace (ColMatrix::num_cols) */
    uint32_t n_traces;       /* STARKPack: traces committed under one tree (>= 1) */
    uint32_t digest_bytes;   /* the hasher: 32 = Blake3_256 (crypto/src/hash/blake/mod.rs:20-59), 24 = Blake3_192 (:68-114: the same
                              * BLAKE3 output truncated to 24 bytes, merge = hash of the 48 bytes of two digests).  HOST arrays
                              * of digests (leaves_out, nodes_out, query outputs) hold digest_bytes per entry, like the
                              * reference's Vec<ByteDigest<N>>; DEVICE arrays (d_leaves, d_nodes of the *_dev forms, resident
                              * handles) are always 32-byte slots with the digest in front and zeros behind; root_out[32]
                              * is the digest zero-padded (Digest::as_bytes, crypto/src/hash/mod.rs:107-113) */
    uint32_t reserved;       /* must be 0 */
    uint8_t domain_offset[16]; /* StarkDomain::offset() as a canonical little-endian integer (7 for f64, 3 for f128) */
} wf_params;

/* ---- context ------------------------------------------------------------------------------------------------- */

/* Creates a context bound to HIP device `device` (twiddle tables, scratch and a stream live in it).
 * One context per GPU; distinct contexts are independent and may be used from different threads at the same time
 * (the reference calls the path from one thread, prover/src/lib.rs:267-268).  A context serves ONE call at a time: a
 * thread that enters while another thread's call is in progress gets WF_ERR_BUSY, nothing is corrupted.  The
 * asynchronous (*_dev) calls share the context's scratch, so calls issued on DIFFERENT streams are ordered on the
 * device by the library: a call first makes its stream wait for everything queued on the stream of the previous call
 * (keep that stream alive until then); calls on one stream run back to back without any host synchronisation.
 * Commitments and FRI provers created on a context use it until they are destroyed: destroy them first, the context
 * last.  (A handle destroyed AFTER its context -- garbage-collected hosts do that at shutdown -- is released safely: the
 * library knows which contexts exist and frees the handle's device buffers without touching the dead context; every other
 * use of such a handle is an error.  Destroying a context twice is ignored.) */
int wf_ctx_create(int device, wf_ctx **out);
void wf_ctx_destroy(wf_ctx *ctx);
/* A context parks the device buffers of destroyed resident commitments (up to 16) for the next commitment of the same
 * shape -- a prover producing proof after proof allocates once.  This returns them to the driver. */
int wf_ctx_release_cached(wf_ctx *ctx);
/* The hasher of the entry points that take no wf_params -- wf_hash_rows, wf_merkle_build, wf_merkle_build_dev and the FRI
 * functions (the reference's `H: ElementHasher` type parameter, one per Prover): 32 = Blake3_256 (default), 24 =
 * Blake3_192.  Layout rules as for wf_params::digest_bytes. */
int wf_ctx_set_digest_bytes(wf_ctx *ctx, uint32_t digest_bytes);
/* Diagnostic (needs no device): the digit passes the commitment path uses for a transform of 2^log2_n rows over
 * n_segments segments (a segment = 8 f64 / 4 f128 base columns); returns the number of passes (<= 4), digits_out[i] =
 * log2 of the tile rows of pass i (the last one is the pass that writes the row-major LDE), or a negative status. */
int wf_plan_digits(uint32_t field, uint32_t log2_n, uint32_t n_segments, uint32_t digits_out[4]);
const char *wf_last_error(void);
/* Number of HIP devices visible (0 if none / no driver). */
int wf_device_count(void);
/* Blocks until all work queued on the context's stream has finished. */
int wf_ctx_synchronize(wf_ctx *ctx);
/* The context's hipStream_t (as void*), for callers that interleave their own work. */
void *wf_ctx_stream(wf_ctx *ctx);

/* Timing of the *_commit_dev calls with HIP events on the call's stream.  on = 1: one event in front of every logical
 * kernel (names "interpolate", "evaluate", "hash_rows", "merkle") and at the end of the call -- 5 events per
 * commitment, ~1 % overhead; on = 2: an event in front of every kernel launch (names such as
 * "evaluate.strided_pass"; ~3 % overhead); on = 0: off.  Events accumulate over calls.  wf_ctx_profile_read waits
 * for the last recorded event, returns the number of (name, milliseconds) pairs written (at most max_entries) and
 * clears the log ("between_calls" = gap to the next call).  Used by bench.py for the roofline figures; off by
 * default. */
int wf_ctx_profile_enable(wf_ctx *ctx, int on);
int wf_ctx_profile_read(wf_ctx *ctx, int max_entries, const char **names, float *ms);

/* Validates a parameter block without touching the device (is_constraint != 0: n_traces must be 1).
 * The preconditions are the reference's assert!s: SURVEY.md §8b "Error convention". */
int wf_params_check(const wf_params *p, int is_constraint);

/* Size helpers (bytes). */
size_t wf_elem_bytes(uint32_t field);
size_t wf_row_width(const wf_params *p);           /* base elements per RowMatrix row */
size_t wf_column_bytes(const wf_params *p);        /* one input/poly column */
size_t wf_lde_bytes(const wf_params *p);           /* one trace's RowMatrix data */
size_t wf_digests_bytes(const wf_params *p);       /* leaves (== nodes) array */

/* ---- the path, host-buffer ("copy-out") form ------------------------------------------------------------------- */

/* Prover::build_trace_commitment (prover/src/lib.rs:615-670):
 *   interpolate every column of every trace (ColMatrix::interpolate_columns, col_matrix.rs:196-206),
 *   evaluate the polynomials over the LDE domain into row-major matrices (RowMatrix::evaluate_polys_over::<8>,
 *   row_matrix.rs:82-98), hash row j of all traces concatenated (commit_to_comb_rows, row_matrix.rs:204-238) and
 *   build the Merkle tree (MerkleTree::new, merkle/mod.rs:117-136).
 * trace_cols[t*n_cols + c] -> column c of trace t (host memory).  Outputs are caller-allocated host buffers:
 *   polys_out[t*n_cols + c] (same shape as the input column; may be NULL to skip the copy-out of all polys),
 *   lde_out[t] (wf_lde_bytes each; the array pointer may be NULL to skip), leaves_out / nodes_out
 *   (wf_digests_bytes each; either may be NULL), root_out (32 bytes, may be NULL). */
int wf_trace_commit(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, void *const *polys_out,
                    void *const *lde_out, uint8_t *leaves_out, uint8_t *nodes_out, uint8_t *root_out);

/* Prover::build_constraint_commitment (prover/src/lib.rs:680-715): composition-polynomial columns (coefficient
 * form, CompositionPoly::data(), constraints/composition_poly.rs:21-41) -> RowMatrix::evaluate_polys_over::<8> ->
 * commit_to_rows (row_matrix.rs:183-203).  p->n_traces must be 1. */
int wf_constraint_commit(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, void *lde_out,
                         uint8_t *leaves_out, uint8_t *nodes_out, uint8_t *root_out);

/* ---- the path, device-resident form ---------------------------------------------------------------------------- */

/* Same computations on caller-owned DEVICE buffers, asynchronous on `stream` (NULL = the context's stream):
 *   d_trace : [n_traces][n_cols] columns, contiguous (wf_column_bytes each)            (read)
 *   d_polys : same shape                                                                (written)
 *   d_lde   : [n_traces] row-major matrices (wf_lde_bytes each)                         (written)
 *   d_leaves, d_nodes : wf_digests_bytes each                                           (written)
 * No host synchronisation happens inside; scratch comes from the context and is reused across calls.  After one call
 * with the same parameters (it allocates the scratch and builds the twiddle tables) the call is a pure sequence of
 * kernel launches and can be captured into a HIP graph on `stream` and replayed (tests/test_gpu_graph.py). */
int wf_trace_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde,
                        void *d_leaves, void *d_nodes, void *stream);
int wf_constraint_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_polys, void *d_lde, void *d_leaves,
                             void *d_nodes, void *stream);

/* ---- the path, coset-sharded over several GPUs (device buffers) ----------------------------------------------------- */

/* One STARKPack commitment (n_traces packed traces, ONE tree) spread over W GPUs by coset: the LDE
 * of size R*blowup is `blowup` independent coset evaluations, and coset c owns exactly the rows j = k*blowup + c.
 * Each rank calls this with its coset range and gets
 *   d_lde_shard    : [n_traces] matrices of (R * coset_count) rows x row_width, local row k*coset_count + (c - begin)
 *   d_leaves_shard : R * coset_count digests in the same local order (each leaf needs only its own row of every trace)
 * d_polys may be NULL.  The ranks then all-gather the leaf shards (the path's one exchange, RCCL), interleave them
 * to natural order (wf_comm_all_gather_leaf_shards) and build the tree with wf_merkle_build_dev -- or use
 * wf_trace_commit_sharded_dev, which also shards the interpolation and the tree. */
int wf_trace_commit_shard_dev(wf_ctx *ctx, const wf_params *p, uint32_t coset_begin, uint32_t coset_count,
                              const void *d_trace, void *d_polys, void *d_lde_shard, void *d_leaves_shard,
                              void *stream);
/* MerkleTree::new on device buffers (leaves -> nodes), asynchronous on `stream`. */
int wf_merkle_build_dev(wf_ctx *ctx, const void *d_leaves, size_t n_leaves, void *d_nodes, void *stream);

/* ---- multi-GPU: one process per GPU, the path's exchanges behind this ABI ------------------------------ */

/* The reference has no distributed code (README.md:43 lists a distributed prover as planned).  Two shardings:
 *  (1) independent proofs, one per GPU (BASELINE configs[3]): no data-path collective; the 32-byte roots are assembled
 *      with ONE all-gather -- wf_comm_all_gather_roots;
 *  (2) one STARKPack commitment (commit_to_comb_rows, row_matrix.rs:204-238) sharded by coset -- wf_trace_commit_sharded_dev.
 * A wf_comm binds a context (= a GPU) to its rank.  Transport: RCCL over xGMI (librccl.so.1 is resolved with dlopen at
 * the first use, so hosts without it can still load this library), or a table of collectives supplied by the caller. */
typedef struct wf_comm wf_comm;
#define WF_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls this and hands the bytes to every rank (over whatever the host uses: a TCP store, MPI). */
int wf_comm_unique_id(uint8_t id_out[WF_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device; collective over all `world` ranks. */
int wf_comm_create(wf_ctx *ctx, const uint8_t id[WF_COMM_ID_BYTES], int rank, int world, wf_comm **out);
/* A host that brings its own fabric code supplies the two collectives the path needs.  Pointers are DEVICE pointers of
 * the calling rank; the transport must order its work after everything queued on `stream` and leave the result visible
 * to work queued on `stream` afterwards (it may simply synchronise).  Return 0 on success.
 *   all_gather: every rank contributes `bytes` at d_send; d_recv receives world * bytes, rank-major
 *   all_to_all: block s of rank r (d_send + s * bytes) lands at d_recv + r * bytes on rank s */
typedef struct wf_transport {
    void *user;
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes, void *stream);
    int (*all_to_all)(void *user, const void *d_send, void *d_recv, size_t bytes, void *stream);
} wf_transport;
int wf_comm_create_with_transport(wf_ctx *ctx, const wf_transport *t, int rank, int world, wf_comm **out);
void wf_comm_destroy(wf_comm *comm);
int wf_comm_rank(const wf_comm *comm);
int wf_comm_world(const wf_comm *comm);
/* Version code of the RCCL library that was loaded (ncclGetVersion), 0 if none could be. */
int wf_comm_rccl_version(void);
/* The file the RCCL symbols were resolved from (dladdr), "" if none was loaded: inside a process that also holds
 * PyTorch's bundled copy this says which of the two the collectives really run on. */
const char *wf_comm_rccl_path(void);
/* ncclAllGather of raw bytes, asynchronous on `stream` (NULL = the context's stream). */
int wf_comm_all_gather(wf_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
/* The one collective of sharding (1): every rank's n_roots roots (32 bytes each, device memory) -> d_all, rank-major. */
int wf_comm_all_gather_roots(wf_comm *comm, const void *d_roots, size_t n_roots, void *d_all, void *stream);
/* Host-blocking helpers for drivers and benchmarks: a barrier over all ranks (through the device: everything queued on
 * the context's stream has finished when it returns) and the maximum of one double over all ranks (in place). */
int wf_comm_barrier(wf_comm *comm);
int wf_comm_max_f64(wf_comm *comm, double *value);
/* One double of every rank to every rank (all_out: wf_comm_world values, rank-major) -- the per-rank step times of a
 * multi-GPU benchmark record.  Host-blocking, under the watchdog. */
int wf_comm_gather_f64(wf_comm *comm, double value, double *all_out);
/* What the transport itself reports about this communicator: transport = 0 (RCCL) or 1 (caller-supplied); for RCCL
 * ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the ncclComm_t in use -- the record of a multi-GPU run shows
 * with these that RCCL really spans `world` ranks, one per device -- for a caller-supplied transport the world, rank
 * and device the communicator was created with.  Any output pointer may be NULL. */
int wf_comm_info(const wf_comm *comm, int *transport, int *nccl_count, int *nccl_user_rank, int *nccl_device);
/* Host-blocking wait for everything queued on `stream` (NULL = the context's stream), collectives included, under the
 * communicator's watchdog: after WF_COMM_TIMEOUT_S seconds (environment, read at wf_comm_create; default 300) without
 * completion -- a peer died or never entered the collective -- the communicator is aborted (ncclCommAbort) and
 * WF_ERR_COMM is returned instead of waiting for ever; every later collective on it fails with WF_ERR_COMM.
 * wf_comm_barrier, wf_comm_max_f64, wf_trace_commit_sharded_resident and wf_sharded_commitment_query wait the same way. */
int wf_comm_stream_wait(wf_comm *comm, void *stream);

/* Partition rules (no device needed; what wf_trace_commit_sharded_dev and the tests use):
 *   proofs : contiguous blocks of proof ids, the first ranks take the remainder
 *   cosets : world (a power of two) must divide blowup; rank r owns cosets [r * blowup / world, (r + 1) * blowup / world)
 *   route  : where LDE row `position` of a sharded commitment lives -- its row on rank row_rank at local row row_local
 *            of d_lde_shard, its leaf on rank tree_rank at index leaf_local of that rank's d_leaves / local tree. */
int wf_shard_proofs(uint32_t n_proofs, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count);
int wf_shard_cosets(uint32_t blowup, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count);
int wf_shard_route(uint32_t log2_lde_rows, uint32_t blowup, uint32_t world, uint64_t position, uint32_t *row_rank,
                   uint64_t *row_local, uint32_t *tree_rank, uint64_t *leaf_local);

/* Replicated-tree form of sharding (2), on top of wf_trace_commit_shard_dev: all-gather of every rank's leaf shard
 * (trace_len * cosets_per_rank digests in (k, local coset) order) into d_leaves in natural row order (n = trace_len *
 * cosets_per_rank * world digests); every rank then builds the whole tree with wf_merkle_build_dev. */
int wf_comm_all_gather_leaf_shards(wf_comm *comm, const void *d_leaves_shard, size_t trace_len, uint32_t cosets_per_rank,
                                   void *d_leaves, void *stream);

/* Sharding (2) with nothing repeated per rank: ONE Prover::build_trace_commitment (prover/src/lib.rs:615-670) of
 * n_traces packed traces spread over the W = wf_comm_world ranks; every rank passes the same parameters and trace.
 *   interpolate : rank r interpolates the segments (groups of 8 f64 / 4 f128 base columns) [r * n_seg / W, ..) and the
 *                 coefficients are all-gathered (when W divides the number of segments; otherwise every rank
 *                 interpolates all columns and nothing is exchanged);
 *   evaluate    : rank r evaluates cosets [r * blowup / W, ..): rows j = k * blowup + c of every trace, and hashes the
 *                 combined rows -- a leaf needs no other rank's data;
 *   exchange    : an all-to-all of digests (R * blowup / W^2 per pair) gives rank r the leaves of the contiguous range
 *                 [r * N / W, (r + 1) * N / W), N = R * blowup; it builds the sub-tree over them;
 *   top         : an all-gather of the W sub-roots (32 bytes each); every rank folds the top log2 W levels.
 * Outputs (device memory of the calling rank):
 *   d_polys     : [n_traces][n_cols] coefficient columns, complete on every rank (may be NULL)
 *   d_lde_shard : [n_traces] matrices of R * blowup / W rows x row_width: local row k * (blowup / W) + (c - first coset)
 *   d_leaves    : N / W digests: leaves r * N / W .. of the tree, natural order
 *   d_nodes     : N / W digests: the sub-tree in MerkleTree::nodes layout (merkle/mod.rs:87-90) -- [1] = sub-root, which is
 *                 node W + r of the whole tree; node i of a level with n >= W nodes is local node i - n - r * n / W + n / W
 *   d_top       : 2 * W digests: nodes 0 .. 2 W - 1 of the whole tree ([0] = zero digest, [1] = the root, [W + s] = the
 *                 sub-root of rank s), identical on every rank
 * wf_shard_route says which rank serves a queried position.  Asynchronous on `stream` unless the transport blocks. */
int wf_trace_commit_sharded_dev(wf_comm *comm, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde_shard,
                                void *d_leaves, void *d_nodes, void *d_top, void *stream);

/* Resident form of the sharded commitment: host columns in (the same on every rank), everything stays in HBM of the
 * rank that owns it.  COLLECTIVE calls (every rank of the communicator, same arguments): ..._resident and ..._query. */
typedef struct wf_sharded_commitment wf_sharded_commitment;
int wf_trace_commit_sharded_resident(wf_comm *comm, const wf_params *p, const void *const *trace_cols,
                                     wf_sharded_commitment **out);
void wf_sharded_commitment_destroy(wf_sharded_commitment *c);
int wf_sharded_commitment_root(const wf_sharded_commitment *c, uint8_t root_out[32]);
/* TraceCommitment::query (prover/src/trace/commitment.rs:87-111) on the sharded commitment: same arguments and outputs as
 * wf_commitment_query, the same on every rank -- each rank gathers the rows of its cosets and the digests of its leaf
 * range / sub-tree at the queried positions, ONE all-gather of a few KiB merges them, the top log2 W levels are local. */
int wf_sharded_commitment_query(wf_sharded_commitment *c, const uint64_t *positions, size_t n, void *rows_out,
                                uint8_t *leaves_out, uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts,
                                size_t *n_vectors, size_t *n_nodes, uint32_t *depth_out);
/* The polynomials (complete on every rank) as a wf_commitment for wf_commitment_evaluate_polys_at -- the out-of-domain
 * frame needs no exchange.  Owned by the sharded commitment; holds no rows (row queries on it fail with WF_ERR_LEAVES). */
struct wf_commitment;
int wf_sharded_commitment_polys(const wf_sharded_commitment *c, const struct wf_commitment **out);

/* ---- the path, resident form: commitment stays in HBM, queries are served from there -------------------------------- */

/* Opaque device-resident commitment: the LDE matrices of all traces, the leaves and the tree nodes (what
 * TraceCommitment / ConstraintCommitment own in the reference: prover/src/trace/commitment.rs:21-26,
 * prover/src/constraints/commitment.rs:21-24) plus the polynomials in column layout.  Avoids copying
 * 0.5-16 GiB over PCIe per commitment; only the <= 255 queried rows and their Merkle paths ever leave the GPU. */
typedef struct wf_commitment wf_commitment;

/* build_trace_commitment with host inputs; polys_out (host, [n_traces*n_cols] pointers) may be NULL. */
int wf_trace_commit_resident(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, void *const *polys_out,
                             wf_commitment **out);
/* A STREAM of proofs from host memory -- STARKPack's workload is many proofs one after the other
 * (examples/src/lib.rs:97-135, winterfell/src/main.rs:105-160): the same build_trace_commitment, but the call returns as
 * soon as the columns are on their way and the kernels are queued; the columns of the next proof travel on the copy stream
 * while the kernels of this one run (two staging buffers in the context).  Pageable host columns: the call returns once
 * they are staged.  Pinned (hipHostMalloc'd / registered) columns: it returns at once and the caller keeps them alive and
 * unchanged until wf_commitment_wait.  The handle can be passed to every wf_commitment_* function straight away (they are
 * ordered behind its kernels on the context's stream); wf_commitment_wait -- or the first wf_commitment_root -- blocks
 * until the root is there and reports a failure of the kernels.  At most 256 commitments in flight (WF_ERR_BUSY). */
int wf_trace_commit_resident_async(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, wf_commitment **out);
int wf_commitment_wait(wf_commitment *c);
/* build_constraint_commitment with host inputs. */
int wf_constraint_commit_resident(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, wf_commitment **out);
/* The constraint side from the combined constraint EVALUATIONS on, without the composition polynomial visiting the host:
 *   - the tail of ConstraintEvaluationTable::into_comb_poly (prover/src/constraints/evaluation_table.rs:166-186):
 *     fft::interpolate_poly_with_offset of every table's combined column over the constraint evaluation domain of
 *     ce_domain_size points with p->domain_offset (the division by the divisors in front of it is AIR-specific: the caller's);
 *   - STARKPack's combination over the n_tables packed traces (prover/src/lib.rs:442-453):
 *     final = comb_0 + sum_{i >= 1} comb_i * final_coeff^i   (final_coeff: one element of E; NULL for one table);
 *   - CompositionPoly::new / segment (constraints/composition_poly.rs:21-41, 86-98): p->n_cols columns of 2^log2_trace_len
 *     coefficients = the first n_cols chunks of the polynomial (ce_domain_size a power of two > trace length);
 *   - build_constraint_commitment (lib.rs:680-715) into a resident handle.
 * combined_evaluations: [n_tables] host arrays of ce_domain_size elements of E (p->ext_degree coordinates); polys_out: NULL or
 * [n_cols] host columns for the composition polynomial's columns.  The handle serves CompositionPoly::evaluate_at
 * (wf_commitment_evaluate_polys_at), the DEEP composition (wf_deep_compose) and ConstraintCommitment::query. */
int wf_constraint_commit_from_evaluations(wf_ctx *ctx, const wf_params *p, const void *const *combined_evaluations, size_t n_tables,
                                          size_t ce_domain_size, const void *final_coeff, void *const *polys_out, wf_commitment **out);
/* The same starting one step earlier, from the constraint evaluation TABLE: all of ConstraintEvaluationTable::into_comb_poly
 * (evaluation_table.rs:166-186) -- acc_column (:335-391) divides every column by its divisor over the constraint evaluation
 * domain x_i = offset * g^i and sums them, get_inv_evaluation (:393-426) supplies 1 / (x^a - b), then the interpolation and the
 * steps above.  A divisor (ConstraintDivisor, air/src/air/divisor.rs:26-29) is (x^a - b) / prod_k (x - e_k) with ONE numerator
 * term (evaluation_table.rs:343 asserts it): a = numerator_degree (a power of two: the trace length for the transition
 * divisor, trace length / stride for assertions), b and the e_k base-field elements in memory representation (exemptions
 * empty for boundary-constraint columns, the last `num_transition_exemptions` trace-domain points for the transition column;
 * at most 8).  columns: [n_columns] host arrays of ce_domain_size elements of E, column j goes with divisors[j].
 * (acc_column indexes the inverses with the position inside its thread's batch; that equals the position in the column
 * whenever the number of distinct inverses divides the batch size -- both are powers of two and batches hold >= 128 rows,
 * so always for the ce_domain_size / trace_length values of a transition divisor.) */
typedef struct wf_divisor {
    uint64_t numerator_degree;
    uint8_t numerator_constant[16];
    const void *exemptions;
    uint32_t n_exemptions;
} wf_divisor;
typedef struct wf_evaluation_table {
    const void *const *columns;
    const wf_divisor *divisors;
    uint32_t n_columns;
} wf_evaluation_table;
int wf_constraint_commit_from_tables(wf_ctx *ctx, const wf_params *p, const wf_evaluation_table *tables, size_t n_tables,
                                     size_t ce_domain_size, const void *final_coeff, void *const *polys_out, wf_commitment **out);
void wf_commitment_destroy(wf_commitment *c);
/* MerkleTree::root (merkle/mod.rs:167) */
int wf_commitment_root(const wf_commitment *c, uint8_t root_out[32]);
/* Number of LDE rows (= leaves), hashed base elements per row over all traces, tree depth. */
int wf_commitment_info(const wf_commitment *c, uint64_t *n_rows, uint64_t *row_elems, uint32_t *depth);
/* Rows at `positions` (TraceCommitment::query / build_segment_queries, trace/commitment.rs:87-111,135-160;
 * ConstraintCommitment::query, constraints/commitment.rs:54-69): rows_out receives n * row_elems base elements,
 * row i = row positions[i] of trace 0 || trace 1 || .. (the "comb_states" that are hashed into leaf positions[i]). */
int wf_commitment_read_rows(const wf_commitment *c, const uint64_t *positions, size_t n, void *rows_out);
/* A contiguous range of rows of ONE trace's matrix as it is stored (RowMatrix::data: row_width base elements per row,
 * padding lanes included; *row_width_out receives that width -- narrow resident constraint commitments keep dense rows) --
 * for hosts that evaluate constraints on the CPU and stream the extended trace in pieces (TraceLde::read_main_trace_frame_into,
 * prover/src/trace/trace_lde.rs:78-98, reads rows i and i + blowup of the same data).  n_rows == 0: only the width. */
int wf_commitment_read_lde(const wf_commitment *c, uint32_t trace, uint64_t row_begin, uint64_t n_rows, void *rows_out,
                           uint64_t *row_width_out);
/* Every row_stride-th row from row_begin on (rows row_begin + k * row_stride, k < n_rows), packed next to each other: the
 * constraint evaluation domain is the LDE domain thinned by lde_blowup / ce_blowup, and the evaluator reads rows
 * step * ce_to_lde_blowup and their successors one trace step later, which are again multiples of that stride
 * (prover/src/trace/trace_lde.rs:78-98) -- a quarter of the matrix for a ce blowup of 2 under an LDE blowup of 8. */
int wf_commitment_read_lde_strided(const wf_commitment *c, uint32_t trace, uint64_t row_begin, uint64_t n_rows, uint64_t row_stride,
                                   void *rows_out, uint64_t *row_width_out);
/* MerkleTree::prove (merkle/mod.rs:192-212): path_out receives (depth + 1) digests: leaf, sibling leaf, siblings
 * bottom-up. */
int wf_commitment_prove(const wf_commitment *c, uint64_t index, uint8_t *path_out);
/* MerkleTree::prove_batch (merkle/mod.rs:222-284), the compressed ("octopus") multi-path proof:
 *   leaves_out  : n digests, leaf of positions[i] at i                          (BatchMerkleProof.leaves)
 *   nodes_out   : the node vectors back to back, at most nodes_capacity digests (BatchMerkleProof.nodes, flattened)
 *   node_counts : length of each vector; *n_vectors of them (at most n)
 *   depth_out   : BatchMerkleProof.depth
 * Errors follow the reference: no positions / more than 255 -> WF_ERR_ARG, out of range or duplicate -> WF_ERR_LEAVES. */
int wf_commitment_prove_batch(const wf_commitment *c, const uint64_t *positions, size_t n, uint8_t *leaves_out,
                              uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts, size_t *n_vectors,
                              size_t *n_nodes, uint32_t *depth_out);
/* TraceCommitment::query / ConstraintCommitment::query (prover/src/trace/commitment.rs:87-111,
 * prover/src/constraints/commitment.rs:54-69) and FriProver::query_layer (fri/src/prover/mod.rs:266-300): the queried
 * rows (as wf_commitment_read_rows) AND their batch proof (as wf_commitment_prove_batch) in one host round trip. */
int wf_commitment_query(const wf_commitment *c, const uint64_t *positions, size_t n, void *rows_out, uint8_t *leaves_out,
                        uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts, size_t *n_vectors, size_t *n_nodes,
                        uint32_t *depth_out);
/* The same for several commitments of ONE context in one host round trip (a proof queries the trace tree, the constraint tree
 * and every FRI layer -- prover/src/lib.rs:593-602, fri/src/prover/mod.rs:244-282 -- ten calls otherwise, ~0.15 ms each):
 * one upload of every position list, the gathers of all commitments, one download, one synchronisation.  Per query the
 * arguments of wf_commitment_query (rows_out may be NULL: proof only); n_vectors, n_nodes and depth are outputs.  The first
 * failing query fails the call with its status and nothing is written. */
typedef struct wf_query {
    const wf_commitment *commitment;
    const uint64_t *positions;
    size_t n;
    void *rows_out;
    uint8_t *leaves_out, *nodes_out;
    size_t nodes_capacity;
    uint32_t *node_counts;
    size_t n_vectors, n_nodes;
    uint32_t depth;
} wf_query;
int wf_commitment_query_many(wf_query *queries, size_t n_queries);

/* ---- FRI layer commitments ---------------------------------------------------------------------- */

/* The commit half of FriProver::build_layer (fri/src/prover/mod.rs:191-203): `evals` = n elements of E (the current
 * layer's evaluations) -> transpose_slice into n/folding rows of `folding` elements (utils/core/src/lib.rs:206-227)
 * -> hash_values (fri/src/utils.rs:41-50) -> MerkleTree::new.  folding in {2,4,8,16} (prover/mod.rs:178-185).
 * Outputs (each may be NULL): the transposed matrix (what FriLayer keeps as `evaluations`), leaves, nodes, root.
 * The root goes to the channel on the host, which draws alpha (out of scope here). */
int wf_fri_layer_commit(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *evals, size_t n,
                        uint32_t folding, void *transposed_out, uint8_t *leaves_out, uint8_t *nodes_out,
                        uint8_t *root_out);
/* folding::apply_drp (fri/src/folding/mod.rs:85-117): rows x folding transposed evaluations, the domain offset and
 * alpha (one element of E, in-memory representation) -> the next layer's `rows` evaluations. */
int wf_fri_apply_drp(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *transposed, size_t rows,
                     uint32_t folding, const uint8_t domain_offset[16], const void *alpha, void *out);
/* Device-buffer forms, asynchronous on `stream` (alpha stays a host pointer: it is a kernel argument). */
int wf_fri_layer_commit_dev(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *d_evals, size_t n,
                            uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes, void *stream);
int wf_fri_apply_drp_dev(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *d_transposed, size_t rows,
                         uint32_t folding, const uint8_t domain_offset[16], const void *alpha, void *d_out,
                         void *stream);

/* Resident form of the whole commit phase: FriProver (fri/src/prover/mod.rs:98-230) with the evaluations, every layer's
 * transposed matrix and every layer's tree kept in HBM.  The channel stays on the host: each layer is two calls with the
 * Fiat-Shamir step in between --
 *     for _ in 0..wf_fri_num_layers(..):   commit_layer -> root;  channel.commit_fri_layer(root);
 *                                          alpha = channel.draw_fri_alpha();  fold(alpha)
 *     set_remainder -> coefficients + their hash_elements commitment                       (prover/mod.rs:172-216)
 * and the query phase (build_proof / query_layer, prover/mod.rs:232-300) reads each layer through the same
 * wf_commitment queries as a trace commitment: positions folded with wf_fri_fold_positions, then
 * wf_commitment_read_rows (the [E; N] evaluations of a position) and wf_commitment_prove_batch. */
typedef struct wf_fri_prover wf_fri_prover;
/* FriOptions::new + FriProver::new (fri/src/options.rs:26-44): folding in {2,4,8,16}, blowup a power of two;
 * domain_offset is FriOptions::domain_offset (the field's GENERATOR in the reference, options.rs:46-55). */
int wf_fri_prover_create(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, uint32_t folding, uint32_t blowup,
                         uint32_t remainder_max_degree, const uint8_t domain_offset[16], wf_fri_prover **out);
void wf_fri_prover_destroy(wf_fri_prover *pr);
/* FriOptions::num_fri_layers (fri/src/options.rs:85-93). */
size_t wf_fri_num_layers(uint32_t folding, uint32_t blowup, uint32_t remainder_max_degree, size_t domain_size);
/* Start of build_layers: the evaluations of the first layer (n elements of E, host memory; _dev: device memory, copied
 * on `stream`).  WF_ERR_ARG while layers of an earlier proof are still held ("a prior proof generation request has not
 * been completed yet", prover/mod.rs:173-176) -- call wf_fri_prover_reset first. */
int wf_fri_prover_begin(wf_fri_prover *pr, const void *evals, size_t n);
int wf_fri_prover_begin_dev(wf_fri_prover *pr, const void *d_evals, size_t n, void *stream);
/* The same start from the DEEP composition polynomial itself (SURVEY.md §8f-2 feeding §8f-1 without a round trip):
 * DeepCompositionPoly::evaluate (prover/src/composer/mod.rs:198-205) = evaluate_poly_with_offset of its n coefficients
 * (elements of E, host memory) over the LDE domain of n * lde_blowup points with the prover's domain offset; the
 * evaluations are produced in HBM and become the first layer (prover/src/lib.rs: fri_prover.build_layers(.., evaluations)). */
int wf_fri_prover_begin_poly(wf_fri_prover *pr, const void *poly, size_t n, size_t lde_blowup);
/* First half of build_layer (prover/mod.rs:191-203): transpose, hash_values, MerkleTree::new; root_out = the layer's
 * root for channel.commit_fri_layer. */
int wf_fri_prover_commit_layer(wf_fri_prover *pr, uint8_t root_out[32]);
/* Second half (prover/mod.rs:205-214): apply_drp with the alpha the channel drew (one element of E, host memory); the
 * committed layer joins the prover's layers, the folded evaluations become the current ones. */
int wf_fri_prover_fold(wf_fri_prover *pr, const void *alpha);
/* set_remainder (prover/mod.rs:218-227): interpolate_poly_with_offset of the current evaluations, the first
 * len / blowup coefficients are the remainder polynomial: remainder_out receives *len_out (<= capacity) elements of E,
 * commitment_out their hash_elements digest (what goes to channel.commit_fri_layer). */
int wf_fri_prover_set_remainder(wf_fri_prover *pr, void *remainder_out, size_t capacity, size_t *len_out,
                                uint8_t commitment_out[32]);
/* Layers built so far and layer i as a resident commitment (rows = positions of the folded domain, one row = the
 * folding * ext_degree base elements of [E; N]; owned by the prover: do not destroy). */
size_t wf_fri_prover_num_layers(const wf_fri_prover *pr);
int wf_fri_prover_layer(const wf_fri_prover *pr, size_t i, const wf_commitment **out);
/* FriProver::reset (prover/mod.rs:150-154): drops the layers (build_proof does this at its end). */
int wf_fri_prover_reset(wf_fri_prover *pr);
/* folding::fold_positions (fri/src/folding/mod.rs:158-175): position % (source_domain_size / folding), first
 * occurrences kept in order; out holds at most n entries. */
int wf_fri_fold_positions(const uint64_t *positions, size_t n, size_t source_domain_size, uint32_t folding,
                          uint64_t *out, size_t *n_out);

/* ---- out-of-domain evaluation ----------------------------------------------------------------------- */

/* ColMatrix::evaluate_columns_at (prover/src/matrix/col_matrix.rs:249-254): every column (n coefficients of
 * ext_degree coordinates) evaluated at one point z of z_ext_degree coordinates (z_ext_degree == ext_degree, or
 * ext_degree == 1): out receives n_cols elements of z's field.  TracePolyTable::get_ood_frame
 * (prover/src/trace/poly_table.rs:67-70) is two calls (z and z*g), CompositionPoly::evaluate_at one. */
int wf_evaluate_columns_at(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *const *poly_cols,
                           size_t n_cols, size_t n, const void *z, uint32_t z_ext_degree, void *out);
/* The same on the polynomials a resident commitment keeps in HBM ([n_traces][n_cols] columns). */
int wf_commitment_evaluate_polys_at(const wf_commitment *c, const void *z, uint32_t z_ext_degree, void *out);
/* The same at up to four points in one call and one host round trip -- TracePolyTable::get_ood_frame(z) is the two points
 * z and z * g (prover/src/trace/poly_table.rs:60-73): points = n_points elements of z_ext_degree coordinates, out receives
 * [n_points][n_traces * n_cols] elements. */
int wf_commitment_evaluate_polys_at_points(const wf_commitment *c, const void *points, uint32_t n_points, uint32_t z_ext_degree,
                                           void *out);

/* ---- DEEP composition polynomial (the caller between the out-of-domain frame and the DEEP LDE + FRI) ------------------ */

/* DeepCompositionPoly::add_trace_polys + add_composition_poly (prover/src/composer/mod.rs:62-193) on the polynomials the
 * resident commitments keep in HBM:
 *     T(x) = sum_i cc_i [ (T_i(x) - T_i(z)) / (x - z) + (T_i(x) - T_i(z g)) / (x - z g) ]   over every column of every trace
 *     H(x) = sum_i cc'_i (H_i(x) - H_i(z)) / (x - z)                                         over the composition columns
 * with g the generator of the trace domain; result T + H: n = trace-length coefficients of E (the top one is zero:
 * degree n - 2, composer/mod.rs:151,192).
 *   trace_commitments   [n_trace_commitments] resident trace commitments of THIS context, all of one field and trace
 *                       length: main segments (ext_degree 1) and auxiliary segments (ext_degree == `ext_degree`), each with
 *                       its n_traces x n_cols columns; `trace_coeffs` holds one element of E per column in that order
 *                       (DeepCompositionCoefficients::traces, air/src/air/coefficients.rs) -- for STARKPack's packed
 *                       traces: handle by handle, trace by trace, column by column.
 *   constraint_commitment  resident constraint commitment (columns of E) or NULL; `constraint_coeffs` one element of E per
 *                       composition column (DeepCompositionCoefficients::constraints).
 *   z                   the out-of-domain point, one element of E (ext_degree coordinates), non-zero.
 * The out-of-domain VALUES the reference's methods are handed (ood_traces_states, ood_evaluations) are not parameters:
 * they are subtracted from coefficient 0 of each accumulator, and syn_div_in_place (math/src/polynom/mod.rs:535-542)
 * never reads coefficient 0 of its dividend for the quotient -- the reference's result does not depend on them.
 *   poly_out            host memory for n elements of E, or NULL.
 *   fri                 NULL, or a FRI prover of this context over (field, ext_degree): the polynomial is evaluated over the
 *                       LDE domain of n * lde_blowup points (DeepCompositionPoly::evaluate, composer/mod.rs:198-205) and
 *                       becomes the prover's first layer without leaving HBM, exactly as wf_fri_prover_begin_poly does
 *                       with a host polynomial.  At least one of poly_out / fri. */
int wf_deep_compose(wf_ctx *ctx, const wf_commitment *const *trace_commitments, size_t n_trace_commitments,
                    const wf_commitment *constraint_commitment, const void *z, uint32_t ext_degree, const void *trace_coeffs,
                    const void *constraint_coeffs, void *poly_out, wf_fri_prover *fri, size_t lde_blowup);

/* ---- building blocks (each mirrors one reference function; host buffers) ---------------------------------------- */

/* fft::evaluate_poly (math/src/fft/mod.rs:85): in place, n elements of ext_degree coordinates, natural order. */
int wf_fft_evaluate_poly(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, void *poly_inout, size_t n);
/* fft::evaluate_poly_with_offset (mod.rs:171): result has n*blowup elements; n a power of two >= 2 and blowup a power of
 * two >= 1, as there (mod.rs:181-201) -- the periodic columns PeriodicValueTable::new evaluates are as short as 2
 * (prover/src/constraints/periodic_table.rs:44-55), LargePolyConstraint::new uses it too (constraints/boundary.rs:426-433). */
int wf_fft_evaluate_poly_with_offset(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *poly, size_t n,
                                     const uint8_t domain_offset[16], size_t blowup, void *result);
/* fft::interpolate_poly (mod.rs:274): in place. */
int wf_fft_interpolate_poly(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, void *evals_inout, size_t n);
/* fft::interpolate_poly_with_offset (mod.rs:362), used by ConstraintEvaluationTable::into_comb_poly
 * (prover/src/constraints/evaluation_table.rs:180-181): in place. */
int wf_fft_interpolate_poly_with_offset(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, void *evals_inout, size_t n,
                                        const uint8_t domain_offset[16]);
/* RowMatrix::evaluate_polys_over::<8> alone (row_matrix.rs:82-98): polys -> one row-major matrix. */
int wf_evaluate_polys_over(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, void *lde_out);
/* ElementHasher::hash_elements for Blake3_256 (crypto/src/hash/blake/mod.rs:46-59) applied to n_rows rows of
 * row_elems base elements each (rows contiguous): digests_out gets n_rows*32 bytes. */
int wf_hash_rows(wf_ctx *ctx, uint32_t field, const void *rows, size_t n_rows, size_t row_elems, uint8_t *digests_out);
/* MerkleTree::new (crypto/src/merkle/mod.rs:117-136): leaves -> nodes (both n_leaves*32 bytes). */
int wf_merkle_build(wf_ctx *ctx, const uint8_t *leaves, size_t n_leaves, uint8_t *nodes_out);

#ifdef __cplusplus
}
#endif
#endif /* WF_LDE_H */
