// Host-side constraint-system surface (product code, C++), mirroring bulletproofs::r1cs as the reference uses it:
//   Variable / LinearCombination                       src/bin/prover.rs:8,245 ; src/conversions.rs:49-64
//   trait ConstraintSystem {multiply, allocate, allocate_multiplier, constrain}   src/cs_buffer.rs:89-113
//   Prover::{new, commit, num_constraints, get_num_multiplications, prove}        src/bin/prover.rs:54,89,92-93
//   Verifier::{new, commit, get_num_vars}                                         src/bin/verifier.rs:51-53,89
// Assembly (witness synthesis, LC evaluation, constraint list) stays on the host exactly as in the reference; only
// commit() and prove() cross the C ABI into the HIP engine.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "scalar.hpp"
#include "merlin.hpp"

namespace bpg {

struct Variable {
    enum Kind : uint32_t { MultiplierLeft = 0, MultiplierRight = 1, MultiplierOutput = 2, Committed = 3, One = 4 };
    Kind kind; uint32_t idx;
    static Variable one() { return Variable{One, 0}; }
    uint32_t packed() const { return (static_cast<uint32_t>(kind) << 29) | idx; }
    static Variable unpack(uint32_t p) { return Variable{static_cast<Kind>(p >> 29), p & 0x1fffffffu}; }
};

// Terms of a linear combination.  Nearly every combination the gadgets build has one to four terms (a MiMC round makes seven of them per
// multiplier pair), so the first four live inside the object and nothing is allocated; longer ones move to the heap.
class TermVec {
public:
    using value_type = std::pair<Variable, Scalar>;
    using iterator = value_type *;
    using const_iterator = const value_type *;
    TermVec() {}
    TermVec(const TermVec &o) { append(o.begin(), o.end()); }
    TermVec(TermVec &&o) noexcept { take(o); }
    TermVec &operator=(const TermVec &o) { if (this != &o) { n_ = 0; append(o.begin(), o.end()); } return *this; }
    TermVec &operator=(TermVec &&o) noexcept { if (this != &o) { n_ = 0; heap_.clear(); take(o); } return *this; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    iterator begin() { return data(); }
    iterator end() { return data() + n_; }
    const_iterator begin() const { return data(); }
    const_iterator end() const { return data() + n_; }
    value_type &operator[](size_t i) { return data()[i]; }
    const value_type &operator[](size_t i) const { return data()[i]; }
    void push_back(const value_type &t) { reserve(n_ + 1); data()[n_++] = t; }
    void emplace_back(const Variable &v, const Scalar &s) { push_back(value_type(v, s)); }
    template <class It> void insert(const_iterator at, It first, It last) {       // appending only (what LinearCombination needs)
        if (at != end()) throw std::logic_error("TermVec::insert: only at end()");
        append(first, last);
    }
    void reserve(size_t need) {
        const size_t cap = heap_.empty() ? INLINE : heap_.size();
        if (need <= cap) return;
        std::vector<value_type> h(std::max(need, 2 * cap));
        std::copy(begin(), end(), h.begin());
        heap_.swap(h);
    }
private:
    static constexpr size_t INLINE = 4;
    template <class It> void append(It first, It last) { for (; first != last; ++first) push_back(*first); }
    void take(TermVec &o) {
        if (!o.heap_.empty()) heap_.swap(o.heap_); else std::copy(o.inl_, o.inl_ + o.n_, inl_);
        n_ = o.n_; o.n_ = 0; o.heap_.clear();
    }
    value_type *data() { return heap_.empty() ? inl_ : heap_.data(); }
    const value_type *data() const { return heap_.empty() ? inl_ : heap_.data(); }
    value_type inl_[INLINE];
    std::vector<value_type> heap_;
    size_t n_ = 0;
};

struct LinearCombination {
    TermVec terms;
    LinearCombination() {}
    LinearCombination(const Variable &v) { terms.emplace_back(v, Scalar::one()); }          // From<Variable>
    LinearCombination(const Scalar &s) { terms.emplace_back(Variable::one(), s); }           // From<Scalar>
    LinearCombination operator+(const LinearCombination &o) const { LinearCombination r = *this; r.terms.insert(r.terms.end(), o.terms.begin(), o.terms.end()); return r; }
    LinearCombination operator-(const LinearCombination &o) const { LinearCombination r = *this; for (auto &t : o.terms) r.terms.emplace_back(t.first, -t.second); return r; }
    LinearCombination operator-() const { LinearCombination r; for (auto &t : terms) r.terms.emplace_back(t.first, -t.second); return r; }
    LinearCombination operator*(const Scalar &s) const { LinearCombination r; for (auto &t : terms) r.terms.emplace_back(t.first, t.second * s); return r; }
};

enum class R1CSError { None = 0, InvalidGeneratorsLength = 1, FormatError = 2, VerificationError = 3, MissingAssignment = 5, GadgetError = 6 };
struct R1CSException : std::runtime_error {
    R1CSError code;
    R1CSException(R1CSError c, const std::string &m) : std::runtime_error(m), code(c) {}
};

struct OptScalar { bool some; Scalar v; OptScalar() : some(false) {} OptScalar(const Scalar &s) : some(true), v(s) {} };
struct MulVars { Variable l, r, o; };

class ConstraintSystem {
public:
    virtual ~ConstraintSystem() {}
    virtual MulVars multiply(LinearCombination left, LinearCombination right) = 0;
    virtual Variable allocate(const OptScalar &assignment) = 0;
    virtual MulVars allocate_multiplier(bool some, const Scalar &l, const Scalar &r) = 0;
    virtual void constrain(const LinearCombination &lc) = 0;
};

// Flattened instance handed to the engine / exported for the oracle (layout of include/bpg.h bpg_r1cs_upload)
struct FlatCircuit {
    uint64_t n = 0, m = 0;
    std::vector<uint8_t> aL, aR, aO;             // n*32 each (empty on the verifier side)
    std::vector<uint64_t> row_ptr{0};
    std::vector<uint32_t> term_var, term_coef;
    std::vector<uint8_t> coef;                   // ncoef*32, deduplicated coefficient table
};

// non-owning view of the same (what crosses the C ABI: bpg_r1cs_instance); witness pointers may be null on the verifier side
struct FlatView {
    uint64_t n = 0, m = 0, q = 0, nnz = 0, ncoef = 0;
    const uint8_t *aL = nullptr, *aR = nullptr, *aO = nullptr;
    const uint64_t *row_ptr = nullptr;
    const uint32_t *term_var = nullptr, *term_coef = nullptr;
    const uint8_t *coef = nullptr;
    FlatView() {}
    explicit FlatView(const FlatCircuit &f)
        : n(f.n), m(f.m), q(f.row_ptr.size() - 1), nnz(f.term_var.size()), ncoef(f.coef.size() / 32),
          aL(f.aL.empty() ? nullptr : f.aL.data()), aR(f.aR.empty() ? nullptr : f.aR.data()), aO(f.aO.empty() ? nullptr : f.aO.data()),
          row_ptr(f.row_ptr.data()), term_var(f.term_var.data()), term_coef(f.term_coef.data()), coef(f.coef.data()) {}
};

// Shared bookkeeping of Prover and Verifier: constraint rows with a coefficient dictionary.
class CircuitCore {
protected:
    struct KeyHash { size_t operator()(const Scalar &s) const { return (size_t)(s.w[0] * 0x9e3779b97f4a7c15ULL ^ s.w[1] ^ (s.w[2] << 1) ^ (s.w[3] >> 3)); } };
    std::vector<uint64_t> row_ptr_{0};
    std::vector<uint32_t> term_var_, term_coef_;
    std::vector<Scalar> coef_;
    std::unordered_map<Scalar, uint32_t, KeyHash> coef_index_;
    std::pair<Scalar, uint32_t> last_[2] = {{Scalar(), UINT32_MAX}, {Scalar(), UINT32_MAX}};
    void push_row(const LinearCombination &lc) {
        for (auto &t : lc.terms) {
            Scalar c = t.second.is_canonical() ? t.second : t.second.reduced();
            uint32_t id;
            if (last_[0].second != UINT32_MAX && c == last_[0].first) id = last_[0].second;           // two-entry memo in front of the hash map
            else if (last_[1].second != UINT32_MAX && c == last_[1].first) { id = last_[1].second; std::swap(last_[0], last_[1]); }
            else {
                auto it = coef_index_.find(c);
                if (it == coef_index_.end()) { id = (uint32_t)coef_.size(); coef_.push_back(c); coef_index_.emplace(c, id); } else id = it->second;
                last_[1] = last_[0]; last_[0] = {c, id};
            }
            term_var_.push_back(t.first.packed()); term_coef_.push_back(id);
        }
        row_ptr_.push_back(term_var_.size());
    }
    void export_rows(FlatCircuit &f) const {
        f.row_ptr = row_ptr_; f.term_var = term_var_; f.term_coef = term_coef_;
        f.coef.resize(coef_.size() * 32);
        for (size_t i = 0; i < coef_.size(); i++) coef_[i].to_bytes(&f.coef[32 * i]);
    }
public:
    size_t num_constraints() const { return row_ptr_.size() - 1; }
};

class Engine;   // engine.hip

class Prover : public ConstraintSystem, public CircuitCore {
public:
    // Prover::new(&pc_gens, &mut transcript): the transcript is borrowed for the prover's lifetime
    Prover(Engine *engine, Transcript *transcript) : engine_(engine), t_(transcript) { t_->r1cs_domain_sep(); }

    // Prover::commit(v, v_blinding) -> (CompressedRistretto, Variable)
    std::pair<std::vector<uint8_t>, Variable> commit(const Scalar &v, const Scalar &v_blinding);
    // batched form of the same call sequence (identical transcript effect, one kernel launch)
    std::vector<Variable> commit_many(const std::vector<Scalar> &v, const std::vector<Scalar> &blind, std::vector<uint8_t> &coms_out);
    // the same for a commitment computed by the caller (no device context needed): registers the variable, appends "V"
    Variable commit_precomputed(const Scalar &v, const Scalar &v_blinding, const uint8_t com[32]);
    // extension (include/bpg.h bpg_prover_defer_commitments): while on, commit / commit_many register their variables and return zero bytes;
    // flush_commitments() makes every pending commitment in ONE kernel launch and appends them to the transcript in the order they were made -
    // the transcript of the per-call path.  prove(), start_blinding() and commit_precomputed() flush first.
    void defer_commitments(bool on) { if (!on) flush_commitments(); deferred_ = on; }
    void flush_commitments();
    // TEST HOOK (include/bpg.h bpg_test_prover_stub_commitments): commitments become 32 hash bytes of (v, blinding) made on the host - NOT group
    // elements - so that the file drivers' parsers and the gadget assembly can be fuzzed under sanitizers without a device; prove() stays refused
    void test_stub_commitments() {
        if (engine_) throw std::invalid_argument("stub commitments are for device-less provers only (a prover with an engine makes real Pedersen commitments)");
        stub_commitments_ = true;
    }
    size_t num_flushed() const { return flushed_; }
    const uint8_t *commitment(size_t i) const { return &V_[32 * i]; }

    MulVars multiply(LinearCombination left, LinearCombination right) override {
        Scalar l = eval(left), r = eval(right), o = l * r;
        uint32_t i = (uint32_t)aL_.size();
        MulVars mv{{Variable::MultiplierLeft, i}, {Variable::MultiplierRight, i}, {Variable::MultiplierOutput, i}};
        aL_.push_back(l); aR_.push_back(r); aO_.push_back(o);
        left.terms.emplace_back(mv.l, -Scalar::one());
        right.terms.emplace_back(mv.r, -Scalar::one());
        constrain(left); constrain(right);
        return mv;
    }
    Variable allocate(const OptScalar &a) override {
        if (!a.some) throw R1CSException(R1CSError::MissingAssignment, "missing assignment");
        if (pending_ < 0) {
            uint32_t i = (uint32_t)aL_.size(); pending_ = i;
            aL_.push_back(a.v); aR_.push_back(Scalar::zero()); aO_.push_back(Scalar::zero());
            return Variable{Variable::MultiplierLeft, i};
        }
        uint32_t i = (uint32_t)pending_; pending_ = -1;
        aR_[i] = a.v; aO_[i] = aL_[i] * aR_[i];
        return Variable{Variable::MultiplierRight, i};
    }
    MulVars allocate_multiplier(bool some, const Scalar &l, const Scalar &r) override {
        if (!some) throw R1CSException(R1CSError::MissingAssignment, "missing assignment");
        uint32_t i = (uint32_t)aL_.size();
        aL_.push_back(l); aR_.push_back(r); aO_.push_back(l * r);
        return MulVars{{Variable::MultiplierLeft, i}, {Variable::MultiplierRight, i}, {Variable::MultiplierOutput, i}};
    }
    void constrain(const LinearCombination &lc) override { push_row(lc); }

    size_t get_num_multiplications() const { return aL_.size(); }
    size_t num_committed() const { return v_.size(); }
    const std::vector<Scalar> &v() const { return v_; }
    const std::vector<Scalar> &v_blinding() const { return vb_; }
    Transcript *transcript() { return t_; }

    Scalar eval(const LinearCombination &lc) const {
        Scalar acc;
        for (auto &t : lc.terms) {
            if (t.first.kind == Variable::One) { acc += t.second; continue; }     // a constant term: c * 1 (round constants, keys) without the product
            const Scalar *x;
            static const Scalar ONE = Scalar::one();
            // a stale or foreign variable (another prover's, a packed value from the C ABI) must not index past the vectors: dalek panics here
            const size_t idx = t.first.idx;
            switch (t.first.kind) {
            case Variable::MultiplierLeft: if (idx >= aL_.size()) throw std::invalid_argument("linear combination refers to an unallocated multiplier"); x = &aL_[idx]; break;
            case Variable::MultiplierRight: if (idx >= aR_.size()) throw std::invalid_argument("linear combination refers to an unallocated multiplier"); x = &aR_[idx]; break;
            case Variable::MultiplierOutput: if (idx >= aO_.size()) throw std::invalid_argument("linear combination refers to an unallocated multiplier"); x = &aO_[idx]; break;
            case Variable::Committed: if (idx >= v_.size()) throw std::invalid_argument("linear combination refers to an uncommitted variable"); x = &v_[idx]; break;
            default: throw std::invalid_argument("linear combination holds an unknown variable kind");
            }
            static const Scalar MINUS_ONE = -Scalar::one();
            if (t.second == ONE) acc += *x;                          // most coefficients of the gadgets are +-1: skip the multiplication
            else if (t.second == MINUS_ONE) acc -= *x;
            else acc += t.second * *x;
        }
        return acc;
    }

    FlatCircuit flatten() const {
        FlatCircuit f; f.n = aL_.size(); f.m = v_.size();
        f.aL.resize(f.n * 32); f.aR.resize(f.n * 32); f.aO.resize(f.n * 32);
        for (size_t i = 0; i < f.n; i++) { red(aL_[i]).to_bytes(&f.aL[32 * i]); red(aR_[i]).to_bytes(&f.aR[32 * i]); red(aO_[i]).to_bytes(&f.aO[32 * i]); }
        export_rows(f);
        return f;
    }

    // Prover::prove(&bp_gens) -> R1CSProof::to_bytes(). rng_seed replaces thread_rng() (32 external bytes).
    std::vector<uint8_t> prove(uint64_t gens_capacity, const uint8_t rng_seed[32], uint32_t flags);
    // extension: start drawing prove()'s blinding scalars now (all commitments made), while the constraints are still being assembled
    void start_blinding(const uint8_t rng_seed[32], uint64_t max_multipliers);

private:
    static Scalar red(const Scalar &s) { return s.is_canonical() ? s : s.reduced(); }
    Engine *engine_;
    Transcript *t_;
    std::vector<Scalar> aL_, aR_, aO_, v_, vb_;
    std::vector<uint8_t> V_;            // the commitments, 32 bytes each (zero until flushed)
    bool deferred_ = false; size_t flushed_ = 0;
    bool stub_commitments_ = false;
    int64_t pending_ = -1;
};

class Verifier : public ConstraintSystem, public CircuitCore {
public:
    explicit Verifier(Transcript *transcript) : t_(transcript) { t_->r1cs_domain_sep(); }
    // Verifier::commit(CompressedRistretto) -> Variable
    Variable commit(const uint8_t com[32]) {
        uint32_t i = (uint32_t)(V_.size() / 32);
        V_.insert(V_.end(), com, com + 32);
        t_->append_point("V", com);
        return Variable{Variable::Committed, i};
    }
    MulVars multiply(LinearCombination left, LinearCombination right) override {
        uint32_t i = (uint32_t)num_vars_++;
        MulVars mv{{Variable::MultiplierLeft, i}, {Variable::MultiplierRight, i}, {Variable::MultiplierOutput, i}};
        left.terms.emplace_back(mv.l, -Scalar::one());
        right.terms.emplace_back(mv.r, -Scalar::one());
        constrain(left); constrain(right);
        return mv;
    }
    Variable allocate(const OptScalar &) override {
        if (pending_ < 0) { pending_ = (int64_t)num_vars_++; return Variable{Variable::MultiplierLeft, (uint32_t)pending_}; }
        uint32_t i = (uint32_t)pending_; pending_ = -1; return Variable{Variable::MultiplierRight, i};
    }
    MulVars allocate_multiplier(bool, const Scalar &, const Scalar &) override {
        uint32_t i = (uint32_t)num_vars_++;
        return MulVars{{Variable::MultiplierLeft, i}, {Variable::MultiplierRight, i}, {Variable::MultiplierOutput, i}};
    }
    void constrain(const LinearCombination &lc) override { push_row(lc); }
    size_t get_num_vars() const { return num_vars_; }
    const std::vector<uint8_t> &commitments() const { return V_; }
    Transcript *transcript() { return t_; }
    FlatCircuit flatten() const { FlatCircuit f; f.n = num_vars_; f.m = V_.size() / 32; export_rows(f); return f; }
private:
    Transcript *t_;
    std::vector<uint8_t> V_;
    size_t num_vars_ = 0;
    int64_t pending_ = -1;
};

}  // namespace bpg
