// winterfell_hip.hpp -- C++ host-side mirror of the reference's interface for the LDE + commitment path, layered on
// the C ABI (wf_lde.h).  The reference is Rust; this image has no Rust toolchain, so the types a winter-prover user
// touches on this path are restated here in C++ with the same names, argument meaning and error behaviour
// (a violated precondition throws where the reference panics / asserts).
//
//   ColMatrix<E>            prover/src/matrix/col_matrix.rs:31-130
//   RowMatrix<E>            prover/src/matrix/row_matrix.rs:31-170
//   MerkleTree              crypto/src/merkle/mod.rs:87-181 (nodes/leaves layout, root, depth, prove, verify)
//   StarkDomain             prover/src/domain.rs:13-155 (the parameters the path reads)
//   CompositionPoly<E>      prover/src/constraints/composition_poly.rs:21-98
//   ConstraintCommitment<E> prover/src/constraints/commitment.rs:21-69
//   Prover                  prover/src/lib.rs:615-715 (build_trace_commitment, build_constraint_commitment)
//   BatchMerkleProof        crypto/src/merkle/proofs.rs:31-38
//   TraceCommitment<E>      prover/src/trace/commitment.rs:21-111 (resident form: LDE + tree stay in HBM; query()
//                           returns the queried rows of all packed traces and the batch Merkle proof)
//
// Header-only; link with -lwf_lde.  Elements are the reference's in-memory representations:
//   F64Element  = Montgomery u64 (math/src/field/f64/mod.rs:48-53)   F128Element = canonical u128
//   QuadExtension<B> / CubeExtension<B> = consecutive base elements (extensions/quadratic.rs:26-28)
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "wf_lde.h"

namespace winterfell {

struct F64Element {
    uint64_t inner;  // Montgomery form, as BaseElement::inner()
    static constexpr uint32_t FIELD = WF_FIELD_F64;
    static constexpr uint32_t EXTENSION_DEGREE = 1;
    typedef F64Element BaseField;
    bool operator==(const F64Element &o) const { return inner == o.inner; }
};
struct F128Element {
    uint64_t lo, hi;  // canonical u128, little endian
    static constexpr uint32_t FIELD = WF_FIELD_F128;
    static constexpr uint32_t EXTENSION_DEGREE = 1;
    typedef F128Element BaseField;
    bool operator==(const F128Element &o) const { return lo == o.lo && hi == o.hi; }
};
template <class B>
struct QuadExtension {
    B c[2];
    static constexpr uint32_t FIELD = B::FIELD;
    static constexpr uint32_t EXTENSION_DEGREE = 2;
    typedef B BaseField;
    bool operator==(const QuadExtension &o) const { return c[0] == o.c[0] && c[1] == o.c[1]; }
};
template <class B>
struct CubeExtension {
    B c[3];
    static constexpr uint32_t FIELD = B::FIELD;
    static constexpr uint32_t EXTENSION_DEGREE = 3;
    typedef B BaseField;
    bool operator==(const CubeExtension &o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2]; }
};

typedef std::array<uint8_t, 32> Digest;  // ByteDigest<32>, crypto/src/hash/mod.rs:84-85

class WfError : public std::runtime_error {
  public:
    WfError(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};
inline void wf_check(int rc) {
    if (rc != 0) throw WfError(rc, std::string("wf_lde: ") + wf_last_error());
}

// ------------------------------------------------------------------------------------------------- ColMatrix
template <class E>
class ColMatrix {
  public:
    // ColMatrix::new (col_matrix.rs:46-60): at least one column, all of equal power-of-two length > 1
    explicit ColMatrix(std::vector<std::vector<E>> columns) : columns_(std::move(columns)) {
        if (columns_.empty()) throw std::invalid_argument("a matrix must contain at least one column");
        const size_t n = columns_[0].size();
        if (n < 2 || (n & (n - 1))) throw std::invalid_argument("number of rows must be a power of two greater than 1");
        for (auto &c : columns_)
            if (c.size() != n) throw std::invalid_argument("all columns must have the same length");
    }
    size_t num_cols() const { return columns_.size(); }
    size_t num_base_cols() const { return columns_.size() * E::EXTENSION_DEGREE; }
    size_t num_rows() const { return columns_[0].size(); }
    const std::vector<E> &get_column(size_t i) const { return columns_.at(i); }
    E get(size_t col, size_t row) const { return columns_.at(col).at(row); }
    const std::vector<std::vector<E>> &columns() const { return columns_; }

  private:
    std::vector<std::vector<E>> columns_;
};

// ------------------------------------------------------------------------------------------------- RowMatrix
template <class E>
class RowMatrix {
  public:
    typedef typename E::BaseField B;
    // the raw constructor the reference lacks
    static RowMatrix from_raw_parts(std::vector<B> data, size_t row_width, size_t elements_per_row) {
        if (row_width == 0 || elements_per_row > row_width || data.size() % row_width)
            throw std::invalid_argument("invalid RowMatrix dimensions");
        RowMatrix m;
        m.data_ = std::move(data);
        m.row_width_ = row_width;
        m.elements_per_row_ = elements_per_row;
        return m;
    }
    size_t num_cols() const { return elements_per_row_ / E::EXTENSION_DEGREE; }  // row_matrix.rs:139-141
    size_t num_rows() const { return data_.size() / row_width_; }                // :144-146
    // RowMatrix::row (row_matrix.rs:160-164)
    const E *row(size_t i) const {
        if (i >= num_rows()) throw std::out_of_range("row index out of bounds");
        return reinterpret_cast<const E *>(data_.data() + i * row_width_);
    }
    E get(size_t col, size_t row_idx) const { return row(row_idx)[col]; }
    const std::vector<B> &data() const { return data_; }
    size_t row_width() const { return row_width_; }
    size_t elements_per_row() const { return elements_per_row_; }

  private:
    std::vector<B> data_;
    size_t row_width_ = 0, elements_per_row_ = 0;
};

// ------------------------------------------------------------------------------------------------- MerkleTree
class MerkleTree {
  public:
    // MerkleTree::from_raw_parts (merkle/mod.rs:149-161)
    static MerkleTree from_raw_parts(std::vector<Digest> nodes, std::vector<Digest> leaves) {
        if (leaves.size() < 2) throw std::invalid_argument("a tree must have at least 2 leaves");
        if (leaves.size() & (leaves.size() - 1)) throw std::invalid_argument("number of leaves must be a power of two");
        if (nodes.size() != leaves.size()) throw std::invalid_argument("nodes and leaves must have the same length");
        MerkleTree t;
        t.nodes_ = std::move(nodes);
        t.leaves_ = std::move(leaves);
        return t;
    }
    // MerkleTree::new (merkle/mod.rs:117-136) on the GPU
    static MerkleTree from_leaves(wf_ctx *ctx, std::vector<Digest> leaves) {
        std::vector<Digest> nodes(leaves.size());
        wf_check(wf_merkle_build(ctx, leaves.empty() ? nullptr : leaves[0].data(), leaves.size(),
                                 nodes.empty() ? nullptr : nodes[0].data()));
        return from_raw_parts(std::move(nodes), std::move(leaves));
    }
    const Digest &root() const { return nodes_[1]; }  // :167
    size_t depth() const {                            // :174-176
        size_t d = 0;
        while (((size_t)1 << d) < leaves_.size()) d++;
        return d;
    }
    const std::vector<Digest> &leaves() const { return leaves_; }
    const std::vector<Digest> &nodes() const { return nodes_; }
    // MerkleTree::prove (merkle/mod.rs:192-212): [leaf, sibling leaf, sibling nodes bottom-up]
    std::vector<Digest> prove(size_t index) const {
        if (index >= leaves_.size()) throw std::out_of_range("leaf index out of bounds");
        std::vector<Digest> proof{leaves_[index], leaves_[index ^ 1]};
        size_t i = (index + nodes_.size()) >> 1;
        while (i > 1) {
            proof.push_back(nodes_[i ^ 1]);
            i >>= 1;
        }
        return proof;
    }

  private:
    std::vector<Digest> nodes_, leaves_;
};

// ------------------------------------------------------------------------------------------------- StarkDomain
class StarkDomain {
  public:
    // the parameters of StarkDomain::new(air) (domain.rs:38-52) that the path reads; offset is the canonical
    // integer of ProofOptions::domain_offset() = B::GENERATOR (air/src/options.rs:199-201)
    StarkDomain(size_t trace_length, size_t blowup, unsigned __int128 domain_offset)
        : trace_length_(trace_length), blowup_(blowup), offset_(domain_offset) {}
    size_t trace_length() const { return trace_length_; }
    size_t trace_to_lde_blowup() const { return blowup_; }
    size_t lde_domain_size() const { return trace_length_ * blowup_; }
    unsigned __int128 offset() const { return offset_; }

  private:
    size_t trace_length_, blowup_;
    unsigned __int128 offset_;
};

// ------------------------------------------------------------------------------------------------- composition poly
template <class E>
class CompositionPoly {
  public:
    // CompositionPoly::new (composition_poly.rs:21-41): split the coefficient vector into `num_columns` columns of
    // `trace_length` coefficients each
    CompositionPoly(const std::vector<E> &coefficients, size_t trace_length, size_t num_columns)
        : data_(split(coefficients, trace_length, num_columns)) {}
    const ColMatrix<E> &data() const { return data_; }
    size_t num_columns() const { return data_.num_cols(); }
    size_t column_len() const { return data_.num_rows(); }

  private:
    static ColMatrix<E> split(const std::vector<E> &c, size_t len, size_t cols) {
        if (c.size() != len * cols) throw std::invalid_argument("coefficient vector length mismatch");
        std::vector<std::vector<E>> out;
        for (size_t i = 0; i < cols; i++) out.emplace_back(c.begin() + i * len, c.begin() + (i + 1) * len);
        return ColMatrix<E>(std::move(out));
    }
    ColMatrix<E> data_;
};

template <class E>
class ConstraintCommitment {
  public:
    // ConstraintCommitment::new (constraints/commitment.rs:29-39)
    ConstraintCommitment(RowMatrix<E> evaluations, MerkleTree tree)
        : evaluations_(std::move(evaluations)), tree_(std::move(tree)) {
        if (evaluations_.num_rows() != tree_.leaves().size())
            throw std::invalid_argument("number of rows in constraint evaluation matrix must be the same as number of leaves");
    }
    const Digest &root() const { return tree_.root(); }
    size_t tree_depth() const { return tree_.depth(); }
    const RowMatrix<E> &evaluations() const { return evaluations_; }
    const MerkleTree &tree() const { return tree_; }

  private:
    RowMatrix<E> evaluations_;
    MerkleTree tree_;
};

// ------------------------------------------------------------------------------------------------- Prover
inline unsigned ilog2_exact(size_t v, const char *what) {
    if (v == 0 || (v & (v - 1))) throw std::invalid_argument(std::string(what) + " must be a power of two");
    unsigned l = 0;
    while (((size_t)1 << l) < v) l++;
    return l;
}

class Prover {
  public:
    explicit Prover(int device = 0) { wf_check(wf_ctx_create(device, &ctx_)); }
    ~Prover() { wf_ctx_destroy(ctx_); }
    Prover(const Prover &) = delete;
    Prover &operator=(const Prover &) = delete;
    wf_ctx *context() const { return ctx_; }

    // Prover::build_trace_commitment (prover/src/lib.rs:615-670)
    template <class E>
    std::tuple<std::vector<RowMatrix<E>>, MerkleTree, std::vector<ColMatrix<E>>> build_trace_commitment(
        const std::vector<const ColMatrix<E> *> &traces, const StarkDomain &domain) const {
        if (traces.empty()) throw std::invalid_argument("at least one trace is required");
        const size_t rows = traces[0]->num_rows(), cols = traces[0]->num_cols();
        wf_params p = params<E>(rows, cols, traces.size(), domain);
        wf_check(wf_params_check(&p, 0));
        std::vector<const void *> in;
        for (auto t : traces) {
            if (t->num_rows() != rows || t->num_cols() != cols) throw std::invalid_argument("all traces must have the same shape");
            for (size_t c = 0; c < cols; c++) in.push_back(t->get_column(c).data());
        }
        const size_t lde_rows = domain.lde_domain_size(), rw = wf_row_width(&p);
        std::vector<std::vector<std::vector<E>>> polys(traces.size(), std::vector<std::vector<E>>(cols, std::vector<E>(rows)));
        std::vector<std::vector<typename E::BaseField>> ldes(traces.size(), std::vector<typename E::BaseField>(lde_rows * rw));
        std::vector<Digest> leaves(lde_rows), nodes(lde_rows);
        std::vector<void *> polys_out, lde_out;
        for (auto &t : polys)
            for (auto &c : t) polys_out.push_back(c.data());
        for (auto &l : ldes) lde_out.push_back(l.data());
        wf_check(wf_trace_commit(ctx_, &p, in.data(), polys_out.data(), lde_out.data(), leaves[0].data(), nodes[0].data(),
                                 nullptr));
        std::vector<RowMatrix<E>> trace_ldes;
        for (auto &l : ldes) trace_ldes.push_back(RowMatrix<E>::from_raw_parts(std::move(l), rw, cols * E::EXTENSION_DEGREE));
        std::vector<ColMatrix<E>> trace_polys;
        for (auto &t : polys) trace_polys.emplace_back(std::move(t));
        return {std::move(trace_ldes), MerkleTree::from_raw_parts(std::move(nodes), std::move(leaves)), std::move(trace_polys)};
    }

    // Prover::build_constraint_commitment (prover/src/lib.rs:680-715)
    template <class E>
    ConstraintCommitment<E> build_constraint_commitment(const CompositionPoly<E> &composition_poly,
                                                        const StarkDomain &domain) const {
        const ColMatrix<E> &data = composition_poly.data();
        wf_params p = params<E>(data.num_rows(), data.num_cols(), 1, domain);
        wf_check(wf_params_check(&p, 1));
        std::vector<const void *> in;
        for (size_t c = 0; c < data.num_cols(); c++) in.push_back(data.get_column(c).data());
        const size_t lde_rows = domain.lde_domain_size(), rw = wf_row_width(&p);
        std::vector<typename E::BaseField> lde(lde_rows * rw);
        std::vector<Digest> leaves(lde_rows), nodes(lde_rows);
        wf_check(wf_constraint_commit(ctx_, &p, in.data(), lde.data(), leaves[0].data(), nodes[0].data(), nullptr));
        return ConstraintCommitment<E>(RowMatrix<E>::from_raw_parts(std::move(lde), rw, data.num_base_cols()),
                                       MerkleTree::from_raw_parts(std::move(nodes), std::move(leaves)));
    }

  private:
    template <class E>
    static wf_params params(size_t rows, size_t cols, size_t n_traces, const StarkDomain &domain) {
        if (rows != domain.trace_length()) throw std::invalid_argument("matrix length does not match the domain");
        wf_params p;
        std::memset(&p, 0, sizeof(p));
        p.field = E::FIELD;
        p.ext_degree = E::EXTENSION_DEGREE;
        p.log2_trace_len = ilog2_exact(rows, "trace length");
        p.log2_blowup = ilog2_exact(domain.trace_to_lde_blowup(), "blowup factor");
        p.n_cols = (uint32_t)cols;
        p.n_traces = (uint32_t)n_traces;
        p.digest_bytes = 32;
        unsigned __int128 off = domain.offset();
        std::memcpy(p.domain_offset, &off, 16);
        return p;
    }
    wf_ctx *ctx_ = nullptr;
};

// ------------------------------------------------------------------------------------------------- resident commitments
struct BatchMerkleProof {  // crypto/src/merkle/proofs.rs:31-38
    std::vector<Digest> leaves;
    std::vector<std::vector<Digest>> nodes;
    uint8_t depth = 0;
};

// TraceCommitment (prover/src/trace/commitment.rs:21-111) whose data stay on the GPU: only queried rows and their
// Merkle proofs are copied to the host.  Built by Prover-like free functions below.
template <class E>
class TraceCommitment {
  public:
    TraceCommitment(wf_commitment *h, size_t n_traces, size_t elements_per_row)
        : h_(h), n_traces_(n_traces), epr_(elements_per_row) {}
    ~TraceCommitment() { wf_commitment_destroy(h_); }
    TraceCommitment(const TraceCommitment &) = delete;
    TraceCommitment &operator=(const TraceCommitment &) = delete;

    Digest main_trace_root() const {  // commitment.rs:137-140
        Digest r;
        wf_check(wf_commitment_root(h_, r.data()));
        return r;
    }
    size_t tree_depth() const {
        uint32_t d = 0;
        wf_check(wf_commitment_info(h_, nullptr, nullptr, &d));
        return d;
    }
    // TraceCommitment::query (commitment.rs:87-111): for every position the row of trace 0 || trace 1 || ..
    // (comb_states), per-trace rows can be sliced out of it, plus the batch proof against the single tree.
    std::pair<std::vector<std::vector<typename E::BaseField>>, BatchMerkleProof> query(
        const std::vector<size_t> &positions) const {
        std::vector<uint64_t> pos(positions.begin(), positions.end());
        const size_t n = pos.size(), row_elems = n_traces_ * epr_;
        std::vector<typename E::BaseField> flat(n * row_elems);
        const size_t depth = tree_depth();
        BatchMerkleProof proof;
        proof.leaves.resize(n);
        std::vector<Digest> nodes(n * (depth + 1));
        std::vector<uint32_t> counts(n);
        size_t n_vec = 0, n_nodes = 0;
        uint32_t d = 0;
        // rows and proof in one host round trip
        // (no positions: the library reports the reference's TooFewLeafIndexes; nothing is dereferenced here)
        wf_check(wf_commitment_query(h_, pos.data(), n, flat.data(), n ? proof.leaves[0].data() : nullptr, n ? nodes[0].data() : nullptr, nodes.size(),
                                     counts.data(), &n_vec, &n_nodes, &d));
        std::vector<std::vector<typename E::BaseField>> rows;
        for (size_t i = 0; i < n; i++) rows.emplace_back(flat.begin() + i * row_elems, flat.begin() + (i + 1) * row_elems);
        size_t k = 0;
        for (size_t i = 0; i < n_vec; i++) {
            proof.nodes.emplace_back(nodes.begin() + k, nodes.begin() + k + counts[i]);
            k += counts[i];
        }
        proof.depth = (uint8_t)d;
        return {std::move(rows), std::move(proof)};
    }

    const wf_commitment *handle() const { return h_; }  // for wf_deep_compose / wf_commitment_evaluate_polys_at

  private:
    wf_commitment *h_;
    size_t n_traces_, epr_;
};

// build_trace_commitment that leaves the commitment on the device (polys are returned to the host as the reference does)
template <class E>
inline std::pair<std::unique_ptr<TraceCommitment<E>>, std::vector<ColMatrix<E>>> build_resident_trace_commitment(
    const Prover &prover, const std::vector<const ColMatrix<E> *> &traces, const StarkDomain &domain) {
    if (traces.empty()) throw std::invalid_argument("at least one trace is required");
    const size_t rows = traces[0]->num_rows(), cols = traces[0]->num_cols();
    wf_params p;
    std::memset(&p, 0, sizeof(p));
    p.field = E::FIELD;
    p.ext_degree = E::EXTENSION_DEGREE;
    p.log2_trace_len = ilog2_exact(rows, "trace length");
    p.log2_blowup = ilog2_exact(domain.trace_to_lde_blowup(), "blowup factor");
    p.n_cols = (uint32_t)cols;
    p.n_traces = (uint32_t)traces.size();
    p.digest_bytes = 32;
    unsigned __int128 off = domain.offset();
    std::memcpy(p.domain_offset, &off, 16);
    std::vector<const void *> in;
    for (auto t : traces)
        for (size_t c = 0; c < cols; c++) in.push_back(t->get_column(c).data());
    std::vector<std::vector<std::vector<E>>> polys(traces.size(), std::vector<std::vector<E>>(cols, std::vector<E>(rows)));
    std::vector<void *> polys_out;
    for (auto &t : polys)
        for (auto &c : t) polys_out.push_back(c.data());
    wf_commitment *h = nullptr;
    wf_check(wf_trace_commit_resident(prover.context(), &p, in.data(), polys_out.data(), &h));
    std::vector<ColMatrix<E>> trace_polys;
    for (auto &t : polys) trace_polys.emplace_back(std::move(t));
    return {std::make_unique<TraceCommitment<E>>(h, traces.size(), cols * E::EXTENSION_DEGREE), std::move(trace_polys)};
}

// build_constraint_commitment (prover/src/lib.rs:680-715) that leaves the commitment on the device: ConstraintCommitment::query
// (constraints/commitment.rs:54-69) is TraceCommitment::query on one matrix, so the same resident class serves it
template <class E>
inline std::unique_ptr<TraceCommitment<E>> build_resident_constraint_commitment(const Prover &prover,
                                                                                const CompositionPoly<E> &composition_poly,
                                                                                const StarkDomain &domain) {
    const size_t cols = composition_poly.num_columns();
    wf_params p;
    std::memset(&p, 0, sizeof(p));
    p.field = E::FIELD;
    p.ext_degree = E::EXTENSION_DEGREE;
    p.log2_trace_len = ilog2_exact(composition_poly.column_len(), "composition column length");
    p.log2_blowup = ilog2_exact(domain.trace_to_lde_blowup(), "blowup factor");
    p.n_cols = (uint32_t)cols;
    p.n_traces = 1;
    p.digest_bytes = 32;
    unsigned __int128 off = domain.offset();
    std::memcpy(p.domain_offset, &off, 16);
    std::vector<const void *> in;
    for (size_t c = 0; c < cols; c++) in.push_back(composition_poly.data().get_column(c).data());
    wf_commitment *h = nullptr;
    wf_check(wf_constraint_commit_resident(prover.context(), &p, in.data(), &h));
    return std::make_unique<TraceCommitment<E>>(h, 1, cols * E::EXTENSION_DEGREE);
}

// The constraint side from the combined constraint evaluations on (ConstraintEvaluationTable::into_comb_poly's
// interpolation, evaluation_table.rs:178-185; STARKPack's final_coeff combination, lib.rs:442-453; into_poly +
// build_constraint_commitment) in one device-resident step.  combined[i]: the combined column of packed trace i over the
// constraint evaluation domain.
template <class E>
inline std::unique_ptr<TraceCommitment<E>> build_resident_constraint_commitment_from_evaluations(
    const Prover &prover, const std::vector<std::vector<E>> &combined, const E &final_coeff, size_t num_cols, const StarkDomain &domain) {
    if (combined.empty()) throw std::invalid_argument("at least one evaluation table is required");
    wf_params p;
    std::memset(&p, 0, sizeof(p));
    p.field = E::FIELD;
    p.ext_degree = E::EXTENSION_DEGREE;
    p.log2_trace_len = ilog2_exact(domain.trace_length(), "trace length");
    p.log2_blowup = ilog2_exact(domain.trace_to_lde_blowup(), "blowup factor");
    p.n_cols = (uint32_t)num_cols;
    p.n_traces = 1;
    p.digest_bytes = 32;
    unsigned __int128 off = domain.offset();
    std::memcpy(p.domain_offset, &off, 16);
    std::vector<const void *> in;
    for (auto &t : combined) {
        if (t.size() != combined[0].size()) throw std::invalid_argument("evaluation tables of different sizes");
        in.push_back(t.data());
    }
    wf_commitment *h = nullptr;
    wf_check(wf_constraint_commit_from_evaluations(prover.context(), &p, in.data(), in.size(), combined[0].size(), &final_coeff, nullptr, &h));
    return std::make_unique<TraceCommitment<E>>(h, 1, num_cols * E::EXTENSION_DEGREE);
}

// ------------------------------------------------------------------------------------------------- FRI prover
struct FriOptions {  // fri/src/options.rs:16-93
    size_t blowup_factor, folding_factor, remainder_max_degree;
    size_t num_fri_layers(size_t domain_size) const {
        return wf_fri_num_layers((uint32_t)folding_factor, (uint32_t)blowup_factor, (uint32_t)remainder_max_degree, domain_size);
    }
};

struct FriProofLayer {  // fri/src/proof.rs: the [E; N] evaluations of the queried positions + their batch proof
    std::vector<uint64_t> positions;           // folded positions this layer was queried at
    std::vector<std::vector<uint64_t>> values; // per position: folding * ext_degree base elements (raw words)
    BatchMerkleProof proof;
};

// FriProver (fri/src/prover/mod.rs:98-300) with evaluations, layers and trees resident in HBM.  `Channel` provides
// commit_fri_layer(const Digest&) and draw_fri_alpha() -> E, like the reference's ProverChannel.
template <class E>
class FriProver {
  public:
    FriProver(wf_ctx *ctx, FriOptions options, unsigned __int128 domain_offset) : options_(options) {
        uint8_t off[16];
        std::memcpy(off, &domain_offset, 16);
        wf_check(wf_fri_prover_create(ctx, E::FIELD, E::EXTENSION_DEGREE, (uint32_t)options.folding_factor,
                                      (uint32_t)options.blowup_factor, (uint32_t)options.remainder_max_degree, off, &h_));
    }
    ~FriProver() { wf_fri_prover_destroy(h_); }
    FriProver(const FriProver &) = delete;
    FriProver &operator=(const FriProver &) = delete;

    size_t num_layers() const { return wf_fri_prover_num_layers(h_); }
    const std::vector<E> &remainder() const { return remainder_; }
    wf_fri_prover *handle() { return h_; }
    void reset() {  // prover/mod.rs:150-154
        wf_check(wf_fri_prover_reset(h_));
        remainder_.clear();
    }

    // build_layers (prover/mod.rs:172-189) + set_remainder (:218-227)
    template <class Channel>
    void build_layers(Channel &channel, const std::vector<E> &evaluations) {
        wf_check(wf_fri_prover_begin(h_, evaluations.data(), evaluations.size()));
        run_layers(channel, evaluations.size());
    }
    // the same, starting from the DEEP composition polynomial (composer/mod.rs:198-205): its evaluations over the LDE
    // domain are produced on the GPU and never visit the host
    template <class Channel>
    void build_layers_from_poly(Channel &channel, const std::vector<E> &coefficients, size_t lde_blowup) {
        wf_check(wf_fri_prover_begin_poly(h_, coefficients.data(), coefficients.size(), lde_blowup));
        run_layers(channel, coefficients.size() * lde_blowup);
    }

    // the same with the polynomial already handed over in HBM (DeepCompositionPoly::compose_into below)
    template <class Channel>
    void build_layers_resident(Channel &channel, size_t n_evaluations) {
        run_layers(channel, n_evaluations);
    }

  private:
    template <class Channel>
    void run_layers(Channel &channel, size_t n_evaluations) {
        domain_size_ = n_evaluations;
        const size_t layers = options_.num_fri_layers(n_evaluations);
        size_t size = n_evaluations;
        for (size_t i = 0; i < layers; i++) {
            Digest root;
            wf_check(wf_fri_prover_commit_layer(h_, root.data()));
            channel.commit_fri_layer(root);
            const E alpha = channel.draw_fri_alpha();
            wf_check(wf_fri_prover_fold(h_, &alpha));
            size /= options_.folding_factor;
        }
        remainder_.resize(size / options_.blowup_factor);
        size_t len = 0;
        Digest commitment;
        wf_check(wf_fri_prover_set_remainder(h_, remainder_.data(), remainder_.size(), &len, commitment.data()));
        remainder_.resize(len);
        channel.commit_fri_layer(commitment);
    }

  public:
    // build_proof (prover/mod.rs:244-282): every layer queried at the folded positions; resets the prover like the reference
    std::pair<std::vector<FriProofLayer>, std::vector<E>> build_proof(const std::vector<size_t> &positions) {
        if (remainder_.empty()) throw std::logic_error("FRI layers have not been built yet");
        // positions of every layer first (host arithmetic), then ONE round trip to the device for all layers
        const size_t n_layers = num_layers();
        std::vector<FriProofLayer> layers(n_layers);
        std::vector<std::vector<uint64_t>> flat(n_layers);
        std::vector<std::vector<Digest>> nodes(n_layers);
        std::vector<std::vector<uint32_t>> counts(n_layers);
        std::vector<size_t> words(n_layers);
        std::vector<wf_query> queries(n_layers);
        std::vector<uint64_t> pos(positions.begin(), positions.end());
        size_t domain = domain_size_;
        for (size_t i = 0; i < n_layers; i++) {
            std::vector<uint64_t> folded(pos.size());
            size_t m = 0;
            wf_check(wf_fri_fold_positions(pos.data(), pos.size(), domain, (uint32_t)options_.folding_factor, folded.data(), &m));
            folded.resize(m);
            const wf_commitment *layer = nullptr;
            wf_check(wf_fri_prover_layer(h_, i, &layer));
            uint64_t n_rows = 0, row_elems = 0;
            uint32_t depth = 0;
            wf_check(wf_commitment_info(layer, &n_rows, &row_elems, &depth));
            words[i] = row_elems * (sizeof(typename E::BaseField) / 8);
            flat[i].resize(m * words[i]);
            layers[i].positions = folded;
            layers[i].proof.leaves.resize(m);
            nodes[i].resize(m * (depth + 1));
            counts[i].resize(m);
            wf_query &q = queries[i];
            std::memset(&q, 0, sizeof(q));
            q.commitment = layer;
            q.positions = layers[i].positions.data();
            q.n = m;
            q.rows_out = flat[i].data();
            q.leaves_out = m ? layers[i].proof.leaves[0].data() : nullptr;
            q.nodes_out = m ? nodes[i][0].data() : nullptr;
            q.nodes_capacity = nodes[i].size();
            q.node_counts = counts[i].data();
            pos = folded;
            domain /= options_.folding_factor;
        }
        if (n_layers) wf_check(wf_commitment_query_many(queries.data(), n_layers));
        for (size_t i = 0; i < n_layers; i++) {
            FriProofLayer &pl = layers[i];
            const size_t m = pl.positions.size();
            for (size_t j = 0; j < m; j++) pl.values.emplace_back(flat[i].begin() + j * words[i], flat[i].begin() + (j + 1) * words[i]);
            size_t k = 0;
            for (size_t v = 0; v < queries[i].n_vectors; v++) {
                pl.proof.nodes.emplace_back(nodes[i].begin() + k, nodes[i].begin() + k + counts[i][v]);
                k += counts[i][v];
            }
            pl.proof.depth = (uint8_t)queries[i].depth;
        }
        std::vector<E> remainder = remainder_;
        reset();
        return {std::move(layers), std::move(remainder)};
    }

  private:
    FriOptions options_;
    wf_fri_prover *h_ = nullptr;
    size_t domain_size_ = 0;
    std::vector<E> remainder_;
};

// ------------------------------------------------------------------------------------------------- DEEP composition
// DeepCompositionCoefficients (air/src/air/coefficients.rs): one element of E per trace column (flattened: commitment by
// commitment, trace by trace, column by column) and per constraint composition column.
template <class E>
struct DeepCompositionCoefficients {
    std::vector<E> traces;
    std::vector<E> constraints;
};

// DeepCompositionPoly (prover/src/composer/mod.rs:16-205) over the polynomials that resident commitments hold in HBM.
// The reference's add_trace_polys / add_composition_poly take the out-of-domain values as well; they only ever touch the
// remainder that syn_div_in_place drops (see wf_deep_compose in wf_lde.h), so this mirror does not ask for them.
template <class E>
class DeepCompositionPoly {
  public:
    DeepCompositionPoly(wf_ctx *ctx, E z, DeepCompositionCoefficients<E> cc) : ctx_(ctx), z_(z), cc_(std::move(cc)) {}

    // add_trace_polys (:62-152) and add_composition_poly (:168-193) in one step; trace_length = TracePolyTable::poly_size
    void add_polys(const std::vector<const wf_commitment *> &trace_commitments, const wf_commitment *constraint_commitment,
                   size_t trace_length) {
        coefficients_.resize(trace_length);
        wf_check(wf_deep_compose(ctx_, trace_commitments.data(), trace_commitments.size(), constraint_commitment, &z_,
                                 E::EXTENSION_DEGREE, cc_.traces.data(), constraint_commitment ? cc_.constraints.data() : nullptr,
                                 coefficients_.data(), nullptr, 0));
    }
    // the same followed by evaluate (:198-205) and FriProver::build_layers: the polynomial and its evaluations stay in HBM
    template <class Channel>
    void add_polys_and_build_fri_layers(const std::vector<const wf_commitment *> &trace_commitments,
                                        const wf_commitment *constraint_commitment, size_t trace_length, size_t lde_blowup,
                                        FriProver<E> &fri, Channel &channel, bool keep_coefficients = false) {
        if (keep_coefficients) coefficients_.resize(trace_length);
        wf_check(wf_deep_compose(ctx_, trace_commitments.data(), trace_commitments.size(), constraint_commitment, &z_,
                                 E::EXTENSION_DEGREE, cc_.traces.data(), constraint_commitment ? cc_.constraints.data() : nullptr,
                                 keep_coefficients ? coefficients_.data() : nullptr, fri.handle(), lde_blowup));
        fri.build_layers_resident(channel, trace_length * lde_blowup);
    }

    size_t poly_size() const { return coefficients_.size(); }  // :40-43
    size_t degree() const {                                    // polynom::degree_of (:45-48)
        static const E zero{};
        for (size_t i = coefficients_.size(); i-- > 0;)
            if (!(coefficients_[i] == zero)) return i;
        return 0;
    }
    const std::vector<E> &coefficients() const { return coefficients_; }

  private:
    wf_ctx *ctx_;
    E z_;
    DeepCompositionCoefficients<E> cc_;
    std::vector<E> coefficients_;
};

}  // namespace winterfell
