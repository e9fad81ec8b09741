// C++ parity test of the host-side mirror (include/winterfell_hip.hpp) -- reads like the reference's own tests:
//   extend_trace_table / commit_trace_table   prover/src/trace/tests.rs:42-128
//   build_fib_trace                            prover/src/tests/mod.rs:17-29
// The oracle (oracle/liboracle.so) is linked only as the checker (hash_elements, merkle, naive eval_many).
// Build + run: tests/test_gpu_cpp_mirror.py (needs a GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/winterfell_hip.hpp"
#include "../../oracle/oracle.h"

using namespace winterfell;
typedef unsigned __int128 u128;

#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static const u128 P128 = (((u128)0xFFFFFFFFFFFFFFFFull) << 64) + (u128)0xFFFFD30000000001ull;
static F128Element f128(u128 v) { return F128Element{(uint64_t)v, (uint64_t)(v >> 64)}; }
static u128 val(F128Element e) { return ((u128)e.hi << 64) | e.lo; }

// prover/src/tests/mod.rs:17-29 (length = 2 * rows)
static ColMatrix<F128Element> build_fib_trace(size_t length) {
    std::vector<F128Element> r1{f128(1)}, r2{f128(1)};
    for (size_t i = 0; i + 1 < length / 2; i++) {
        u128 a = val(r1[i]), b = val(r2[i]);
        r1.push_back(f128((a + b) % P128));
        r2.push_back(f128((a + 2 * b) % P128));
    }
    return ColMatrix<F128Element>({r1, r2});
}

static void extend_and_commit_trace_table() {
    const size_t trace_length = 8, blowup = 8;  // MockAir::with_trace_length(8): ProofOptions blowup 8
    ColMatrix<F128Element> trace = build_fib_trace(trace_length * 2);
    StarkDomain domain(trace_length, blowup, 3);  // f128 GENERATOR
    Prover prover(0);
    auto [ldes, tree, polys] = prover.build_trace_commitment<F128Element>({&trace}, domain);

    EXPECT(trace.get_column(0)[7] == f128(610) && trace.get_column(1)[7] == f128(987));  // new_trace_table
    EXPECT(ldes.size() == 1 && polys.size() == 1);
    EXPECT(ldes[0].num_cols() == 2 && ldes[0].num_rows() == 64 && ldes[0].row_width() == 8);

    // trace polynomials evaluate to the Fibonacci trace over the trace domain (tests.rs:62-80)
    u128 root_b;
    orc_f128_get_root_of_unity(3, &root_b);
    std::vector<u128> dom(trace_length), out(trace_length);
    u128 acc = 1, one = 1;
    for (size_t i = 0; i < trace_length; i++) {
        dom[i] = acc;
        orc_f128_mul(&acc, &root_b, &acc);
    }
    (void)one;
    for (size_t c = 0; c < 2; c++) {
        orc_f128_eval_many(reinterpret_cast<const u128 *>(polys[0].get_column(c).data()), trace_length, dom.data(),
                           trace_length, out.data());
        for (size_t i = 0; i < trace_length; i++) EXPECT(out[i] == val(trace.get_column(c)[i]));
    }
    // LDE columns are the polynomials on the shifted domain (tests.rs:82-92, via direct evaluation)
    u128 g, x = 3;
    orc_f128_get_root_of_unity(6, &g);
    for (size_t j = 0; j < 64; j++) {
        for (size_t c = 0; c < 2; c++) {
            u128 y;
            orc_f128_eval_many(reinterpret_cast<const u128 *>(polys[0].get_column(c).data()), trace_length, &x, 1, &y);
            EXPECT(y == val(ldes[0].get(c, j)));
        }
        for (size_t l = 2; l < 8; l++) EXPECT(val(ldes[0].row(j)[l]) == 0);  // padding lanes
        orc_f128_mul(&x, &g, &x);
    }
    // commit_trace_table (tests.rs:95-128): tree over manually hashed rows
    std::vector<Digest> hashed(64), nodes(64);
    for (size_t i = 0; i < 64; i++) {
        F128Element state[2] = {ldes[0].get(0, i), ldes[0].get(1, i)};
        orc_hash_elements(ORC_FIELD_F128, state, 2, hashed[i].data());
    }
    EXPECT(orc_build_merkle_nodes(hashed[0].data(), 64, nodes[0].data(), 1) == 0);
    EXPECT(nodes[1] == tree.root());
    EXPECT(hashed == tree.leaves() && nodes == tree.nodes());
    EXPECT(tree.depth() == 6);
    // MerkleTree::prove / verify round trip (merkle/tests.rs:94-131 shape)
    for (size_t idx : {0u, 1u, 37u, 63u}) {
        auto proof = tree.prove(idx);
        EXPECT(proof.size() == 7);
        Digest v;
        size_t r = idx & 1;
        orc_merge(proof[r].data(), proof[1 - r].data(), v.data());
        size_t index = (idx + 64) >> 1;
        for (size_t k = 2; k < proof.size(); k++) {
            if ((index & 1) == 0) orc_merge(v.data(), proof[k].data(), v.data());
            else orc_merge(proof[k].data(), v.data(), v.data());
            index >>= 1;
        }
        EXPECT(v == tree.root());
    }
    std::printf("extend_and_commit_trace_table ok\n");
}

static void starkpack_two_traces_f64() {
    // two packed traces under one tree: leaf = hash(row of trace 0 || row of trace 1) (row_matrix.rs:204-238)
    const size_t n = 32, blowup = 4, cols = 3;
    std::vector<ColMatrix<F64Element>> traces;
    uint64_t s = 12345;
    for (int t = 0; t < 2; t++) {
        std::vector<std::vector<F64Element>> c(cols, std::vector<F64Element>(n));
        for (auto &col : c)
            for (auto &e : col) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                e.inner = orc_f64_new(s >> 3);
            }
        traces.emplace_back(c);
    }
    StarkDomain domain(n, blowup, 7);
    Prover prover(0);
    auto [ldes, tree, polys] = prover.build_trace_commitment<F64Element>({&traces[0], &traces[1]}, domain);
    EXPECT(ldes.size() == 2 && polys.size() == 2);
    std::vector<Digest> hashed(n * blowup), nodes(n * blowup);
    for (size_t i = 0; i < n * blowup; i++) {
        uint64_t comb[2 * cols];
        for (int t = 0; t < 2; t++)
            for (size_t c = 0; c < cols; c++) comb[t * cols + c] = ldes[t].get(c, i).inner;
        orc_hash_elements(ORC_FIELD_F64, comb, 2 * cols, hashed[i].data());
    }
    orc_build_merkle_nodes(hashed[0].data(), hashed.size(), nodes[0].data(), 1);
    EXPECT(nodes[1] == tree.root());
    // the whole thing against the oracle's path
    std::vector<const void *> in;
    std::vector<std::vector<uint64_t>> po(2 * cols, std::vector<uint64_t>(n)), lo(2, std::vector<uint64_t>(n * blowup * 8));
    std::vector<void *> pout, lout;
    for (auto &t : traces)
        for (size_t c = 0; c < cols; c++) in.push_back(t.get_column(c).data());
    for (auto &v : po) pout.push_back(v.data());
    for (auto &v : lo) lout.push_back(v.data());
    std::vector<Digest> ol(n * blowup), on(n * blowup);
    uint8_t off[16] = {7};
    EXPECT(orc_build_trace_commitment(ORC_FIELD_F64, 1, 5, 2, cols, 2, off, in.data(), pout.data(), lout.data(),
                                      ol[0].data(), on[0].data(), 1) == 0);
    EXPECT(on[1] == tree.root());
    for (int t = 0; t < 2; t++)
        EXPECT(std::memcmp(lo[t].data(), ldes[t].data().data(), lo[t].size() * 8) == 0);
    std::printf("starkpack_two_traces_f64 ok\n");
}

static void constraint_commitment_quadratic() {
    typedef QuadExtension<F64Element> E;
    const size_t n = 64, blowup = 8, cols = 2;
    std::vector<E> coeffs(n * cols);
    uint64_t s = 99;
    for (auto &e : coeffs)
        for (auto &c : e.c) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            c.inner = orc_f64_new(s >> 2);
        }
    CompositionPoly<E> poly(coeffs, n, cols);
    EXPECT(poly.num_columns() == cols && poly.column_len() == n);
    StarkDomain domain(n, blowup, 7);
    Prover prover(0);
    ConstraintCommitment<E> cc = prover.build_constraint_commitment(poly, domain);
    EXPECT(cc.tree_depth() == 9 && cc.evaluations().num_cols() == cols && cc.evaluations().row_width() == 8);
    std::vector<const void *> in{poly.data().get_column(0).data(), poly.data().get_column(1).data()};
    std::vector<uint64_t> lde(n * blowup * 8);
    std::vector<Digest> ol(n * blowup), on(n * blowup);
    uint8_t off[16] = {7};
    EXPECT(orc_build_constraint_commitment(ORC_FIELD_F64, 2, 6, 3, cols, off, in.data(), lde.data(), ol[0].data(),
                                           on[0].data(), 1) == 0);
    EXPECT(on[1] == cc.root());
    EXPECT(std::memcmp(lde.data(), cc.evaluations().data().data(), lde.size() * 8) == 0);
    std::printf("constraint_commitment_quadratic ok\n");
}

static void errors_like_the_reference() {
    Prover prover(0);
    bool threw = false;
    try {  // trace shorter than 8 rows (air/src/air/trace_info.rs:35)
        ColMatrix<F64Element> t({std::vector<F64Element>(4)});
        StarkDomain d(4, 8, 7);
        prover.build_trace_commitment<F64Element>({&t}, d);
    } catch (const WfError &e) {
        threw = e.code == WF_ERR_TRACE_LENGTH;
    }
    EXPECT(threw);
    threw = false;
    try {
        MerkleTree::from_leaves(prover.context(), std::vector<Digest>(1));  // merkle/mod.rs:118-120
    } catch (const WfError &e) {
        threw = e.code == WF_ERR_LEAVES;
    }
    EXPECT(threw);
    threw = false;
    try {
        ColMatrix<F64Element> bad({std::vector<F64Element>(6)});  // not a power of two (col_matrix.rs:50-54)
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw);
    std::printf("errors_like_the_reference ok\n");
}

static void resident_commitment_query() {
    // TraceCommitment::query on a commitment that never leaves the GPU (trace/commitment.rs:87-111)
    const size_t n = 64, blowup = 8, cols = 4;
    std::vector<ColMatrix<F64Element>> traces;
    uint64_t s = 777;
    for (int t = 0; t < 2; t++) {
        std::vector<std::vector<F64Element>> c(cols, std::vector<F64Element>(n));
        for (auto &col : c)
            for (auto &e : col) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                e.inner = orc_f64_new(s >> 5);
            }
        traces.emplace_back(c);
    }
    StarkDomain domain(n, blowup, 7);
    Prover prover(0);
    auto [full_ldes, full_tree, full_polys] = prover.build_trace_commitment<F64Element>({&traces[0], &traces[1]}, domain);
    auto [com, polys] = build_resident_trace_commitment<F64Element>(prover, {&traces[0], &traces[1]}, domain);
    EXPECT(com->main_trace_root() == full_tree.root());
    EXPECT(com->tree_depth() == 9);
    EXPECT(polys[1].get_column(2) == full_polys[1].get_column(2));
    std::vector<size_t> positions{5, 4, 511, 100, 0};
    auto [rows, proof] = com->query(positions);
    EXPECT(rows.size() == positions.size() && proof.leaves.size() == positions.size() && proof.depth == 9);
    for (size_t i = 0; i < positions.size(); i++) {
        EXPECT(rows[i].size() == 2 * cols);
        for (int t = 0; t < 2; t++)
            for (size_t c = 0; c < cols; c++) EXPECT(rows[i][t * cols + c] == full_ldes[t].get(c, positions[i]));
        Digest h;
        orc_hash_elements(ORC_FIELD_F64, rows[i].data(), rows[i].size(), h.data());
        EXPECT(h == proof.leaves[i] && h == full_tree.leaves()[positions[i]]);
    }
    // positions 4 and 5 are siblings: their pair needs no leaf-level node (merkle/mod.rs:238-252)
    EXPECT(proof.nodes.size() == 4);
    std::printf("resident_commitment_query ok\n");
}

// fri/src/prover/tests.rs:22-69 (fri_prove_verify): evaluations of a polynomial with coefficients 0..trace_length-1 over
// the LDE domain, folding factor 4, max remainder degree 7 -- here the commit phase and the query phase of the resident
// FriProver against the oracle's layer-by-layer restatement.  The channel derives alpha from the layer root.
struct TestChannel {
    std::vector<Digest> layer_commitments;
    void commit_fri_layer(const Digest &d) { layer_commitments.push_back(d); }
    F64Element draw_fri_alpha() {
        Digest seed;
        orc_merge_with_int(layer_commitments.back().data(), layer_commitments.size(), seed.data());
        uint64_t v;
        std::memcpy(&v, seed.data(), 8);
        return F64Element{orc_f64_new(v >> 2)};
    }
};

static void fri_prover_resident() {
    const size_t trace_length = 1 << 8, blowup = 8, folding = 4, max_rem = 7, n = trace_length * blowup;
    std::vector<uint64_t> tw(n / 2);
    EXPECT(orc_f64_get_twiddles(tw.data(), n, 0) == 0);
    std::vector<F64Element> evaluations(n, F64Element{0});
    for (size_t i = 0; i < trace_length; i++) evaluations[i].inner = orc_f64_new(i);  // build_evaluations
    orc_f64_evaluate_poly(reinterpret_cast<uint64_t *>(evaluations.data()), n, 1, tw.data());

    Prover prover(0);
    FriOptions options{blowup, folding, max_rem};
    FriProver<F64Element> fri(prover.context(), options, 7);
    TestChannel channel;
    fri.build_layers(channel, evaluations);
    EXPECT(fri.num_layers() == options.num_fri_layers(n) && fri.num_layers() == 3);
    EXPECT(channel.layer_commitments.size() == fri.num_layers() + 1);

    // oracle: the same chain on the host
    std::vector<uint64_t> cur(n);
    for (size_t i = 0; i < n; i++) cur[i] = evaluations[i].inner;
    TestChannel want;
    size_t size = n;
    uint8_t off[16] = {7};
    std::vector<std::vector<uint64_t>> transposed;
    std::vector<std::vector<Digest>> trees;
    for (size_t l = 0; l < 3; l++) {
        std::vector<uint64_t> tr(size);
        orc_transpose_slice(ORC_FIELD_F64, cur.data(), size, 1, folding, tr.data());
        const size_t rows = size / folding;
        std::vector<Digest> leaves(rows), nodes(rows);
        for (size_t i = 0; i < rows; i++) orc_hash_elements(ORC_FIELD_F64, &tr[i * folding], folding, leaves[i].data());
        EXPECT(orc_build_merkle_nodes(leaves[0].data(), rows, nodes[0].data(), 1) == 0);
        EXPECT(nodes[1] == channel.layer_commitments[l]);
        want.commit_fri_layer(nodes[1]);
        const F64Element alpha = want.draw_fri_alpha();
        std::vector<uint64_t> next(rows);
        orc_apply_drp(ORC_FIELD_F64, tr.data(), rows, 1, folding, off, &alpha.inner, next.data(), 1);
        transposed.push_back(tr);
        trees.push_back(nodes);
        cur = next;
        size = rows;
    }
    std::vector<uint64_t> itw(size / 2);
    EXPECT(orc_f64_get_twiddles(itw.data(), size, 1) == 0);
    orc_f64_interpolate_poly_with_offset(cur.data(), size, 1, itw.data(), orc_f64_new(7));
    EXPECT(fri.remainder().size() == size / blowup);
    for (size_t i = 0; i < size / blowup; i++) EXPECT(fri.remainder()[i].inner == cur[i]);
    Digest rc;
    orc_hash_elements(ORC_FIELD_F64, cur.data(), size / blowup, rc.data());
    EXPECT(rc == channel.layer_commitments.back());

    // query phase (build_proof): values of every layer at the folded positions + verifiable batch proofs' leaves
    auto [layers, remainder] = fri.build_proof({5, 77, 1029, 2000, 1500, 77 + 512});
    EXPECT(layers.size() == 3 && remainder.size() == size / blowup && fri.num_layers() == 0);
    size_t domain = n;
    for (size_t l = 0; l < 3; l++) {
        const size_t target = domain / folding;
        for (size_t j = 0; j < layers[l].positions.size(); j++) {
            const uint64_t pos = layers[l].positions[j];
            EXPECT(pos < target);
            for (size_t k = 0; k < folding; k++) EXPECT(layers[l].values[j][k] == transposed[l][pos * folding + k]);
            Digest h;
            orc_hash_elements(ORC_FIELD_F64, layers[l].values[j].data(), folding, h.data());
            EXPECT(h == layers[l].proof.leaves[j]);
        }
        EXPECT(layers[l].proof.depth == ilog2_exact(target, "layer size"));
        domain = target;
    }
    EXPECT(layers[0].positions.size() == 4);  // 1029 folds onto 5 and 77 + 512 onto 77 in the 512-row layer

    // the DEEP-polynomial entry: coefficients in, coset LDE on the device; first layer root against the oracle
    std::vector<F64Element> coeffs(evaluations.begin(), evaluations.begin() + trace_length);
    for (size_t i = 0; i < trace_length; i++) coeffs[i].inner = orc_f64_new(3 * i + 1);
    TestChannel ch2;
    fri.build_layers_from_poly(ch2, coeffs, blowup);
    std::vector<uint64_t> ttw(trace_length / 2), lde(n), tr0(n);
    EXPECT(orc_f64_get_twiddles(ttw.data(), trace_length, 0) == 0);
    orc_f64_evaluate_poly_with_offset(reinterpret_cast<const uint64_t *>(coeffs.data()), trace_length, 1, ttw.data(),
                                      orc_f64_new(7), blowup, lde.data());
    orc_transpose_slice(ORC_FIELD_F64, lde.data(), n, 1, folding, tr0.data());
    std::vector<Digest> l0(n / folding), n0(n / folding);
    for (size_t i = 0; i < n / folding; i++) orc_hash_elements(ORC_FIELD_F64, &tr0[i * folding], folding, l0[i].data());
    EXPECT(orc_build_merkle_nodes(l0[0].data(), n / folding, n0[0].data(), 1) == 0);
    EXPECT(n0[1] == ch2.layer_commitments[0]);
    std::printf("fri_prover_resident ok\n");
}

struct QuadChannel {
    std::vector<Digest> layer_commitments;
    void commit_fri_layer(const Digest &d) { layer_commitments.push_back(d); }
    QuadExtension<F64Element> draw_fri_alpha() {
        Digest seed;
        orc_merge_with_int(layer_commitments.back().data(), layer_commitments.size(), seed.data());
        uint64_t v[2];
        std::memcpy(v, seed.data(), 16);
        return QuadExtension<F64Element>{{F64Element{orc_f64_new(v[0] >> 2)}, F64Element{orc_f64_new(v[1] >> 2)}}};
    }
};

// DeepCompositionPoly over two packed f64 traces and a quadratic-extension constraint commitment, everything resident;
// checked against the literal CPU restatement of composer/mod.rs (with the out-of-domain values it is handed there)
static void deep_composition_resident() {
    typedef QuadExtension<F64Element> E;
    const size_t trace_length = 1 << 9, blowup = 8, n_cols = 3, n_traces = 2, n_cons = 2;
    Prover prover(0);
    StarkDomain domain(trace_length, blowup, 7);
    uint64_t seed = 12345;
    auto next = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return orc_f64_new(seed >> 3); };
    std::vector<ColMatrix<F64Element>> traces;
    for (size_t t = 0; t < n_traces; t++) {
        std::vector<std::vector<F64Element>> cols(n_cols, std::vector<F64Element>(trace_length));
        for (auto &c : cols)
            for (auto &v : c) v.inner = next();
        traces.emplace_back(std::move(cols));
    }
    auto [commitment, trace_polys] = build_resident_trace_commitment<F64Element>(prover, {&traces[0], &traces[1]}, domain);
    // constraint composition columns over E, committed resident through the C ABI
    std::vector<E> composition(n_cons * trace_length);
    for (auto &v : composition) v = E{{F64Element{next()}, F64Element{next()}}};
    CompositionPoly<E> composition_poly(composition, trace_length, n_cons);
    auto constraint_commitment = build_resident_constraint_commitment<E>(prover, composition_poly, domain);
    const wf_commitment *cons_commitment = constraint_commitment->handle();
    const void *cons_ptrs[n_cons] = {composition_poly.data().get_column(0).data(), composition_poly.data().get_column(1).data()};

    const E z{{F64Element{next()}, F64Element{next()}}};
    DeepCompositionCoefficients<E> cc;
    for (size_t i = 0; i < n_traces * n_cols; i++) cc.traces.push_back(E{{F64Element{next()}, F64Element{next()}}});
    for (size_t i = 0; i < n_cons; i++) cc.constraints.push_back(E{{F64Element{next()}, F64Element{next()}}});
    DeepCompositionPoly<E> deep(prover.context(), z, cc);
    deep.add_polys({commitment->handle()}, cons_commitment, trace_length);
    EXPECT(deep.poly_size() == trace_length && deep.degree() == trace_length - 2);  // composer/mod.rs:151

    // the reference's route on the CPU: out-of-domain frame first, then the composer
    const uint64_t g = orc_f64_get_root_of_unity(9);
    const uint64_t zg[2] = {orc_f64_mul(z.c[0].inner, g), orc_f64_mul(z.c[1].inner, g)};
    std::vector<const void *> cols;
    std::vector<size_t> col_ext, per_table(n_traces, n_cols);
    std::vector<uint64_t> ood_z, ood_zg, ood_c;
    for (size_t t = 0; t < n_traces; t++)
        for (size_t c = 0; c < n_cols; c++) {
            const uint64_t *poly = reinterpret_cast<const uint64_t *>(trace_polys[t].get_column(c).data());
            cols.push_back(poly);
            col_ext.push_back(1);
            uint64_t v[2];
            orc_eval_column_at(ORC_FIELD_F64, poly, trace_length, 1, &z, 2, v);
            ood_z.insert(ood_z.end(), v, v + 2);
            orc_eval_column_at(ORC_FIELD_F64, poly, trace_length, 1, zg, 2, v);
            ood_zg.insert(ood_zg.end(), v, v + 2);
        }
    for (size_t c = 0; c < n_cons; c++) {
        uint64_t v[2];
        orc_eval_column_at(ORC_FIELD_F64, cons_ptrs[c], trace_length, 2, &z, 2, v);
        ood_c.insert(ood_c.end(), v, v + 2);
    }
    std::vector<uint64_t> want(trace_length * 2);
    orc_deep_compose(ORC_FIELD_F64, 2, trace_length, n_traces, per_table.data(), cols.data(), col_ext.data(), ood_z.data(),
                     ood_zg.data(), cc.traces.data(), n_cons, cons_ptrs, ood_c.data(), cc.constraints.data(), &z, want.data());
    EXPECT(std::memcmp(want.data(), deep.coefficients().data(), want.size() * 8) == 0);

    // straight into FRI: first layer root = the one a prover started from the host coefficients commits to
    FriOptions options{blowup, 4, 7};
    FriProver<E> fri_a(prover.context(), options, 7), fri_b(prover.context(), options, 7);
    QuadChannel ch_a, ch_b;
    DeepCompositionPoly<E> deep2(prover.context(), z, cc);
    deep2.add_polys_and_build_fri_layers({commitment->handle()}, cons_commitment, trace_length, blowup, fri_a, ch_a);
    fri_b.build_layers_from_poly(ch_b, deep.coefficients(), blowup);
    EXPECT(ch_a.layer_commitments.size() == ch_b.layer_commitments.size() && ch_a.layer_commitments.size() >= 2);
    for (size_t i = 0; i < ch_a.layer_commitments.size(); i++) EXPECT(ch_a.layer_commitments[i] == ch_b.layer_commitments[i]);
    EXPECT(fri_a.remainder().size() == fri_b.remainder().size());
    for (size_t i = 0; i < fri_a.remainder().size(); i++) EXPECT(fri_a.remainder()[i] == fri_b.remainder()[i]);
    // the same constraint commitment reached from combined EVALUATIONS over a constraint evaluation domain of 4 R points:
    // evaluate the composition polynomial there on the CPU, hand the evaluations over, compare the roots
    {
        const size_t ce = 4 * trace_length;
        std::vector<uint64_t> tw(ce / 2), coeffs(ce * 2, 0), evals(ce * 2);
        EXPECT(orc_f64_get_twiddles(tw.data(), ce, 0) == 0);
        std::memcpy(coeffs.data(), composition.data(), n_cons * trace_length * 16);  // the columns are the polynomial's chunks
        orc_f64_evaluate_poly_with_offset(coeffs.data(), ce, 2, tw.data(), orc_f64_new(7), 1, evals.data());
        std::vector<std::vector<E>> combined(1, std::vector<E>(ce));
        std::memcpy(combined[0].data(), evals.data(), ce * 16);
        auto from_evals = build_resident_constraint_commitment_from_evaluations<E>(prover, combined, z, n_cons, domain);
        EXPECT(from_evals->main_trace_root() == constraint_commitment->main_trace_root());
    }
    std::printf("deep_composition_resident ok\n");
}

int main() {
    fri_prover_resident();
    deep_composition_resident();
    extend_and_commit_trace_table();
    resident_commitment_query();
    starkpack_two_traces_f64();
    constraint_commitment_quadratic();
    errors_like_the_reference();
    std::printf("ALL OK\n");
    return 0;
}
