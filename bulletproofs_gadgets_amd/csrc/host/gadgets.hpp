// Host-side gadget layer (product code, C++): the reference's Gadget trait and the three gadgets the benchmark
// configurations need, with the same names, argument meaning and constraint order, so that multiplier / constraint
// indices (and therefore proofs) line up with the reference's assembly:
//   conversions        reference src/conversions.rs:6-76
//   range_proof        src/utils.rs:5-35
//   trait Gadget       src/gadget.rs:6-59
//   BoundsCheck        src/bounds_check/bounds_check_gadget.rs:8-63
//   mimc_hash (native) src/mimc_hash/mimc.rs:7-97
//   MimcHash256        src/mimc_hash/mimc_hash_gadget.rs:7-150
//   MerkleTree256      src/merkle_tree/merkle_tree_gadget.rs:14-113
//   commitments        src/commitments.rs:8-47
//   Equality           src/equality/equality_gadget.rs:5-40
//   Inequality         src/inequality/inequality_gadget.rs:5-113
//   LessThan           src/less_than/less_than_gadget.rs:6-86
//   SetMembership      src/set_membership/set_membership_gadget.rs:5-132
//   OpBuffer / or_conjunction   src/cs_buffer.rs:5-113, src/or/or_conjunction.rs:4-67
#pragma once
#include <memory>
#include "r1cs.hpp"

namespace bpg {

typedef std::vector<uint8_t> Bytes;

// ------------------------------------------------------------------------------------------ conversions.rs
inline std::vector<Scalar> le_to_scalars(Bytes b) {                      // :6-23
    if (b.size() % 32 != 0) b.resize(b.size() + (32 - b.size() % 32), 0);
    std::vector<Scalar> out;
    for (size_t i = 0; i < b.size(); i += 32) out.push_back(Scalar::from_bits(&b[i]));
    return out;
}
inline std::vector<Scalar> be_to_scalars(const Bytes &b) { return le_to_scalars(Bytes(b.rbegin(), b.rend())); }   // :26-30
inline Scalar le_to_scalar(Bytes b) {                                    // :33-46
    if (b.size() > 32) throw std::invalid_argument("the given vector is longer than 32 bytes");
    b.resize(32, 0);
    return Scalar::from_bits(b.data());
}
inline Scalar be_to_scalar(const Bytes &b) { return le_to_scalar(Bytes(b.rbegin(), b.rend())); }                   // :49-53
inline Bytes scalar_to_be(const Scalar &s) { Bytes b(s.as_bytes(), s.as_bytes() + 32); return Bytes(b.rbegin(), b.rend()); }   // :72-76
inline std::vector<LinearCombination> vars_to_lc(const std::vector<Variable> &v) { return std::vector<LinearCombination>(v.begin(), v.end()); }

// ------------------------------------------------------------------------------------------ utils.rs:5-35
inline void range_proof(ConstraintSystem &cs, LinearCombination x, uint8_t n, const OptScalar &x_assignment) {
    Scalar exp_2 = Scalar::one();
    for (uint8_t i = 0; i < n; i++) {
        Scalar l, r;
        if (x_assignment.some) {
            uint8_t off = i / 8;
            uint8_t bit = (x_assignment.v.as_bytes()[off] >> (i - off * 8)) & 1u;
            l = Scalar::from_u64(1u - bit); r = Scalar::from_u64(bit);
        }
        MulVars mv = cs.allocate_multiplier(x_assignment.some, l, r);
        cs.constrain(LinearCombination(mv.o));                                                   // a * b = 0
        cs.constrain(LinearCombination(mv.l) + (LinearCombination(mv.r) - LinearCombination(Scalar::one())));   // a = 1 - b
        x = x - LinearCombination(mv.r) * exp_2;
        exp_2 = exp_2 + exp_2;
    }
    cs.constrain(x);
}

// ------------------------------------------------------------------------------------------ gadget.rs:6-59
typedef std::vector<std::pair<OptScalar, Variable>> Derived;

class Gadget {
public:
    virtual ~Gadget() {}
    virtual std::vector<Scalar> preprocess(const std::vector<Scalar> &witnesses) const = 0;
    virtual void assemble(ConstraintSystem &cs, const std::vector<Variable> &witnesses, const Derived &derived) const = 0;
    // setup(): one Pedersen commitment per derived scalar; blindings supplied by the caller (thread_rng upstream)
    std::pair<std::vector<uint8_t>, Derived> setup(Prover &prover, const std::vector<Scalar> &witnesses, const std::vector<Scalar> &blindings) const {
        std::vector<Scalar> d = preprocess(witnesses);
        if (blindings.size() < d.size()) throw std::invalid_argument("setup: not enough blinding factors");
        std::vector<uint8_t> coms;
        std::vector<Variable> vars = prover.commit_many(d, std::vector<Scalar>(blindings.begin(), blindings.begin() + d.size()), coms);
        Derived out;
        for (size_t i = 0; i < d.size(); i++) out.emplace_back(OptScalar(d[i]), vars[i]);
        return {coms, out};
    }
    void prove(ConstraintSystem &cs, const std::vector<Variable> &commitment_vars, const Derived &derived) const { assemble(cs, commitment_vars, derived); }
    void verify(ConstraintSystem &cs, const std::vector<Variable> &witnesses, const std::vector<Variable> &derived) const {
        Derived d; for (auto &v : derived) d.emplace_back(OptScalar(), v);
        assemble(cs, witnesses, d);
    }
};

// ------------------------------------------------------------------------------------------ bounds_check_gadget.rs
class BoundsCheck : public Gadget {
public:
    BoundsCheck(const Bytes &min, const Bytes &max) : min_(be_to_scalar(min)), max_(be_to_scalar(max)), n_((uint8_t)(max.size() * 8)) {}   // :54-63
    std::vector<Scalar> preprocess(const std::vector<Scalar> &w) const override { return {w.at(0) - min_, max_ - w.at(0)}; }             // :14-21
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &, const Derived &d) const override {                                 // :23-47
        LinearCombination a_lc(d.at(0).second), b_lc(d.at(1).second);
        cs.constrain((a_lc + b_lc) - LinearCombination(max_ - min_));
        range_proof(cs, a_lc, n_, d[0].first);
        range_proof(cs, b_lc, n_, d[1].first);
    }
private:
    Scalar min_, max_; uint8_t n_;
};

// ------------------------------------------------------------------------------------------ MiMC
const std::vector<Scalar> &mimc_round_constants();       // 486 constants, n = 769 (mimc_consts data)

inline Scalar mimc_encryption(const Scalar &p, const Scalar &k, const std::vector<Scalar> &c) {   // mimc.rs:7-23
    Scalar state = p;
    for (size_t i = 0; i < c.size(); i++) { Scalar t = state + (k + c[i]); state = (t * t) * t; }
    return state + k;
}
inline Scalar mimc_sponge_1(const std::vector<Scalar> &pre, const std::vector<Scalar> &c) {      // mimc.rs:26-40
    Scalar state;
    for (auto &b : pre) { state += b; state = mimc_encryption(state, Scalar::zero(), c); }
    return state;
}
// PKCS#7 to 32 bytes on the little-endian bytes of the last block with trailing zeros stripped (mimc.rs:77-97)
inline bool mimc_pad_last(const Scalar &last, Scalar &padded) {
    Bytes le(last.as_bytes(), last.as_bytes() + 32);
    while (!le.empty() && le.back() == 0) le.pop_back();
    if (le.size() < 32) { uint8_t k = (uint8_t)(32 - le.size()); le.resize(32, k); padded = le_to_scalar(le); return true; }
    padded = le_to_scalar(Bytes(32, 32)); return false;
}
inline Scalar mimc_hash(const Bytes &preimage) {                                                  // mimc.rs:61-75
    std::vector<Scalar> pre = be_to_scalars(preimage);
    if (pre.empty()) throw std::invalid_argument("mimc_hash: empty preimage");
    Scalar padded;
    if (mimc_pad_last(pre.back(), padded)) pre.pop_back();
    pre.push_back(padded);
    return mimc_sponge_1(pre, mimc_round_constants());
}

class MimcHash256 : public Gadget {
public:
    static constexpr size_t ROUNDS = 486;
    MimcHash256() : image_(LinearCombination(Scalar::zero())) {}                                   // init()   :58-63
    explicit MimcHash256(const LinearCombination &image) : image_(image) {}                        // new()    :65-70
    std::vector<Scalar> preprocess(const std::vector<Scalar> &w) const override {                  // :15-37
        if (w.empty()) throw R1CSException(R1CSError::GadgetError, "MimcHash256: empty witness");        // the reference indexes [len - 1] and panics
        Scalar padded;
        if (mimc_pad_last(w.back(), padded)) return {padded, padded - w.back()};
        return {padded};
    }
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &w, const Derived &d) const override {   // :39-50
        std::vector<Variable> coms = pad(cs, w, d);
        LinearCombination h = mimc_sponge(cs, vars_to_lc(coms));
        cs.constrain(h - image_);
    }
    LinearCombination mimc_sponge(ConstraintSystem &cs, const std::vector<LinearCombination> &pre) const {   // :108-122
        LinearCombination key_zero(Scalar::zero()), state(Scalar::zero());
        for (auto &v : pre) { state = state + v; state = mimc_encryption_lc(cs, state, key_zero); }
        return state;
    }
private:
    std::vector<Variable> pad(ConstraintSystem &cs, const std::vector<Variable> &w, const Derived &d) const {   // :81-106
        std::vector<Variable> coms = w;
        Variable padded_block = d.at(0).second;
        if (d.size() == 2) {
            LinearCombination last(coms.back()); coms.pop_back();
            cs.constrain((last + LinearCombination(d[1].second)) - LinearCombination(padded_block));
        }
        coms.push_back(padded_block);
        return coms;
    }
    LinearCombination mimc_encryption_lc(ConstraintSystem &cs, LinearCombination p, const LinearCombination &k) const {   // :124-150
        const std::vector<Scalar> &rc = mimc_round_constants();
        for (size_t i = 0; i < ROUNDS; i++) {
            LinearCombination t = (p + k) + LinearCombination(rc[i]);
            MulVars sq = cs.multiply(t, t);
            MulVars cube = cs.multiply(LinearCombination(sq.o), LinearCombination(sq.l));
            p = LinearCombination(cube.o);
        }
        return p + k;
    }
    LinearCombination image_;
};

// ------------------------------------------------------------------------------------------ merkle_tree_gadget.rs
struct Pattern {
    enum Kind { Hash, W, I } kind;
    std::shared_ptr<Pattern> left, right;
    static std::shared_ptr<Pattern> leaf(Kind k) { auto p = std::make_shared<Pattern>(); p->kind = k; return p; }
    static std::shared_ptr<Pattern> hash(std::shared_ptr<Pattern> l, std::shared_ptr<Pattern> r) { auto p = std::make_shared<Pattern>(); p->kind = Hash; p->left = l; p->right = r; return p; }
    // textual form "(W (I W))" with W / I leaves, as in the .gadgets tree syntax (gadget_grammar.lalrpop:54-79)
    // Nesting is bounded (MAX_DEPTH levels: a path through 2^64 leaves is no tree anybody proves): the parser, the assembly and the destructor recurse once per
    // level, and the text comes from a file - ten thousand opening brackets were a stack overflow (found by the fuzzer of tests/hostcheck, round 5).
    static constexpr uint32_t MAX_DEPTH = 64;
    static std::shared_ptr<Pattern> parse(const std::string &s) { size_t pos = 0; auto p = parse_at(s, pos, 0); skip(s, pos); if (pos != s.size()) throw std::invalid_argument("pattern: trailing input"); return p; }
private:
    static void skip(const std::string &s, size_t &pos) { while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t')) pos++; }
    static std::shared_ptr<Pattern> parse_at(const std::string &s, size_t &pos, uint32_t depth) {
        if (depth > MAX_DEPTH) throw std::invalid_argument("pattern: nested deeper than 64 levels");
        skip(s, pos);
        if (pos >= s.size()) throw std::invalid_argument("pattern: unexpected end");
        if (s[pos] == 'W') { pos++; return leaf(W); }
        if (s[pos] == 'I') { pos++; return leaf(I); }
        if (s[pos] != '(') throw std::invalid_argument("pattern: expected '(', 'W' or 'I'");
        pos++;
        auto l = parse_at(s, pos, depth + 1); auto r = parse_at(s, pos, depth + 1);
        skip(s, pos);
        if (pos >= s.size() || s[pos] != ')') throw std::invalid_argument("pattern: expected ')'");
        pos++;
        return hash(l, r);
    }
};

class MerkleTree256 : public Gadget {
public:
    MerkleTree256(const LinearCombination &root, const std::vector<LinearCombination> &instance_vars,
                  const std::vector<LinearCombination> &witness_vars, std::shared_ptr<Pattern> pattern)
        : root_(root), inst_(instance_vars), wit_(witness_vars), pattern_(pattern) {}
    std::vector<Scalar> preprocess(const std::vector<Scalar> &) const override { return {}; }     // :40-42
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &, const Derived &) const override {   // :44-56
        size_t wi = 0, ii = 0;
        LinearCombination h = parse(cs, wi, ii, *pattern_);
        cs.constrain(h - root_);
    }
private:
    // :75-107. Children are evaluated left to right; a W / I child consumes the next witness / instance value.
    LinearCombination parse(ConstraintSystem &cs, size_t &wi, size_t &ii, const Pattern &p) const {
        std::vector<LinearCombination> pre;
        if (p.kind == Pattern::Hash) { pre.push_back(child(cs, wi, ii, *p.left)); pre.push_back(child(cs, wi, ii, *p.right)); }
        else pre.push_back(child(cs, wi, ii, p));
        return gadget_.mimc_sponge(cs, pre);
    }
    LinearCombination child(ConstraintSystem &cs, size_t &wi, size_t &ii, const Pattern &p) const {
        if (p.kind == Pattern::W) { if (wi >= wit_.size()) throw std::invalid_argument("too few variables provided to satisfy the given pattern"); return wit_[wi++]; }
        if (p.kind == Pattern::I) { if (ii >= inst_.size()) throw std::invalid_argument("too few variables provided to satisfy the given pattern"); return inst_[ii++]; }
        return parse(cs, wi, ii, p);
    }
    LinearCombination root_;
    std::vector<LinearCombination> inst_, wit_;
    std::shared_ptr<Pattern> pattern_;
    MimcHash256 gadget_;
};

// ------------------------------------------------------------------------------------------ equality_gadget.rs
// LEFT = RIGHT limb by limb; LEFT is a witness (variables), RIGHT witness or instance (linear combinations)
class Equality : public Gadget {
public:
    explicit Equality(const std::vector<LinearCombination> &right_hand) : right_(right_hand) {}
    std::vector<Scalar> preprocess(const std::vector<Scalar> &) const override { return {}; }
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &left, const Derived &) const override {
        if (right_.size() != left.size()) { cs.constrain(LinearCombination(Scalar::one())); return; }      // unsatisfiable: 1 = 0
        for (size_t i = 0; i < left.size(); i++) cs.constrain(right_[i] - LinearCombination(left[i]));
    }
private:
    std::vector<LinearCombination> right_;
};

// ------------------------------------------------------------------------------------------ inequality_gadget.rs
// LEFT != RIGHT: per limb delta = |left - right| (byte-wise comparison), delta * delta^-1 in {0,1}, and the sum of those has an inverse
class Inequality : public Gadget {
public:
    Inequality(const std::vector<LinearCombination> &right_hand, bool has_assignment, const std::vector<Scalar> &right_assignment)
        : right_(right_hand), has_(has_assignment), right_val_(right_assignment) {}
    static bool compare(const Scalar &l, const Scalar &r) {                                 // :103-113, true when l >= r as 32-byte LE integers
        for (int i = 31; i >= 0; i--) { uint8_t a = l.as_bytes()[i], b = r.as_bytes()[i]; if (a > b) return true; if (a < b) return false; }
        return true;
    }
    std::vector<Scalar> preprocess(const std::vector<Scalar> &left) const override {        // :12-45
        if (!has_) throw std::invalid_argument("missing right hand assignment");
        std::vector<Scalar> d; Scalar sum;
        for (size_t i = 0; i < left.size(); i++) {
            const Scalar r = i < right_val_.size() ? right_val_[i] : Scalar::zero();
            const Scalar delta = compare(left[i], r) ? left[i] - r : r - left[i];
            d.push_back(delta);
            if (delta == Scalar::zero()) d.push_back(Scalar::zero());
            else { Scalar inv = delta.invert(); d.push_back(inv); sum = sum + delta * inv; }
        }
        d.push_back(sum.invert());
        return d;
    }
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &left, const Derived &d) const override {   // :47-92
        if (right_.size() != left.size()) { cs.constrain(LinearCombination(Scalar::zero())); return; }
        LinearCombination sum(Scalar::zero());
        for (size_t i = 0; i < left.size(); i++) {
            const LinearCombination l(left[i]), delta(d.at(2 * i).second), delta_inv(d.at(2 * i + 1).second);
            MulVars z = cs.multiply((l - right_[i]) - delta, (right_[i] - l) - delta);      // (l - r - delta)(r - l - delta) = 0
            cs.constrain(LinearCombination(z.o));
            MulVars zo = cs.multiply(delta, delta_inv);
            sum = sum + LinearCombination(zo.o);
        }
        MulVars one = cs.multiply(sum, LinearCombination(d.back().second));
        cs.constrain(LinearCombination(Scalar::one()) - LinearCombination(one.o));
    }
private:
    std::vector<LinearCombination> right_; bool has_; std::vector<Scalar> right_val_;
};

// ------------------------------------------------------------------------------------------ less_than_gadget.rs
// LEFT < RIGHT with both in [0, 2^126): delta = right - left is range-checked and shown non-zero
class LessThan : public Gadget {
public:
    LessThan(const LinearCombination &left, const OptScalar &left_assignment, const LinearCombination &right, const OptScalar &right_assignment)
        : left_(left), right_(right), la_(left_assignment), ra_(right_assignment) {}
    std::vector<Scalar> preprocess(const std::vector<Scalar> &) const override {            // :16-35
        if (!la_.some || !ra_.some) throw std::invalid_argument("missing right hand assignment");
        const Scalar delta = ra_.v - la_.v;
        return {delta, delta == Scalar::zero() ? Scalar::zero() : delta.invert()};
    }
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &, const Derived &d) const override {   // :37-66
        const LinearCombination delta(d.at(0).second), delta_inv(d.at(1).second);
        range_proof(cs, left_, 126, la_);
        range_proof(cs, right_, 126, ra_);
        range_proof(cs, delta, 126, d[0].first);
        MulVars one = cs.multiply(delta, delta_inv);
        cs.constrain(LinearCombination(Scalar::one()) - LinearCombination(one.o));
        cs.constrain((right_ - left_) - delta);
    }
private:
    LinearCombination left_, right_; OptScalar la_, ra_;
};

// ------------------------------------------------------------------------------------------ set_membership_gadget.rs
// value is an element of (witness set || instance set): one-hot selector (derived), bits, sum 1, <selector, set> = value
class SetMembership : public Gadget {
public:
    SetMembership(const LinearCombination &value, const OptScalar &value_assignment, const std::vector<LinearCombination> &instance_vars,
                  bool has_instance_assignments, const std::vector<Scalar> &instance_assignments)
        : value_(value), va_(value_assignment), inst_(instance_vars), has_(has_instance_assignments), inst_val_(instance_assignments) {}
    std::vector<Scalar> preprocess(const std::vector<Scalar> &witnesses) const override {   // :13-34
        if (!va_.some) throw std::invalid_argument("missing value assignment");
        if (!has_) throw std::invalid_argument("missing instance vars assignments");
        std::vector<Scalar> d;
        for (auto &e : witnesses) d.push_back(e == va_.v ? Scalar::one() : Scalar::zero());
        for (auto &e : inst_val_) d.push_back(e == va_.v ? Scalar::one() : Scalar::zero());
        return d;
    }
    void assemble(ConstraintSystem &cs, const std::vector<Variable> &witnesses, const Derived &d) const override {   // :36-61
        std::vector<LinearCombination> one_hot;
        for (auto &b : d) {
            LinearCombination bit(b.second);
            MulVars z = cs.multiply(LinearCombination(Scalar::one()) - bit, bit);          // is_bit :98-110
            cs.constrain(LinearCombination(z.o));
            one_hot.push_back(bit);
        }
        LinearCombination sum(Scalar::zero());                                              // one_hot_vector :80-95
        for (auto &b : one_hot) sum = sum + b;
        cs.constrain(LinearCombination(Scalar::one()) - sum);
        std::vector<LinearCombination> set;
        for (auto &w : witnesses) set.push_back(LinearCombination(w));
        for (auto &e : inst_) set.push_back(e);
        if (one_hot.size() != set.size()) { cs.constrain(LinearCombination(Scalar::one())); return; }   // hadamard_product :112-131
        LinearCombination prod(Scalar::zero());
        for (size_t i = 0; i < set.size(); i++) { MulVars m = cs.multiply(one_hot[i], set[i]); prod = prod + LinearCombination(m.o); }
        cs.constrain(value_ - prod);
    }
private:
    LinearCombination value_; OptScalar va_; std::vector<LinearCombination> inst_; bool has_; std::vector<Scalar> inst_val_;
};

// ------------------------------------------------------------------------------------------ cs_buffer.rs / or_conjunction.rs
// Recording constraint system for OR blocks.  The reference wraps a throw-away shadow Prover/Verifier whose only observable
// effect is the numbering of the multiplier variables it hands out; here that numbering is a counter that starts at the
// parent's current multiplier count (what the reference reaches by replaying `initialization` into the shadow).
struct BufferedOp {
    enum Kind { Multiply, AllocateMultiplier, Constrain } kind;
    LinearCombination a, b;            // Multiply: (left, right); Constrain: a
    bool some = false; Scalar l, r;    // AllocateMultiplier
};
class OpBuffer : public ConstraintSystem {
public:
    OpBuffer(uint64_t first_multiplier, bool prover_side) : next_(first_multiplier), prover_(prover_side) {}
    MulVars multiply(LinearCombination left, LinearCombination right) override {
        BufferedOp op; op.kind = BufferedOp::Multiply; op.a = std::move(left); op.b = std::move(right); ops_.push_back(std::move(op));
        return vars(next_++);
    }
    Variable allocate(const OptScalar &) override { throw R1CSException(R1CSError::GadgetError, "call to unimplemented method allocate"); }   // cs_buffer.rs:99-101
    MulVars allocate_multiplier(bool some, const Scalar &l, const Scalar &r) override {
        if (prover_ && !some) throw R1CSException(R1CSError::MissingAssignment, "missing assignment");   // cs_buffer.rs:103-104
        BufferedOp op; op.kind = BufferedOp::AllocateMultiplier; op.some = some && prover_; op.l = l; op.r = r; ops_.push_back(std::move(op));
        return vars(next_++);
    }
    void constrain(const LinearCombination &lc) override { BufferedOp op; op.kind = BufferedOp::Constrain; op.a = lc; ops_.push_back(std::move(op)); }
    void rewind() { cached_.push_back(std::move(ops_)); ops_.clear(); }                              // cs_buffer.rs:75-78
    const std::vector<std::vector<BufferedOp>> &cache() const { return cached_; }
    uint64_t next_multiplier() const { return next_; }
private:
    static MulVars vars(uint64_t i) { uint32_t k = (uint32_t)i; return MulVars{{Variable::MultiplierLeft, k}, {Variable::MultiplierRight, k}, {Variable::MultiplierOutput, k}}; }
    uint64_t next_; bool prover_;
    std::vector<BufferedOp> ops_;
    std::vector<std::vector<BufferedOp>> cached_;
};

// or(main, buffer): replay every clause's multipliers into main, then constrain the product of one constraint per clause to
// zero for every element of the Cartesian product (last clause varying fastest, as the reference's fold does)
inline void or_conjunction(ConstraintSystem &main, const OpBuffer &buffer) {
    std::vector<std::vector<const LinearCombination *>> sets;
    for (const auto &clause : buffer.cache()) {
        std::vector<const LinearCombination *> cons;
        for (const BufferedOp &op : clause) {
            switch (op.kind) {
            case BufferedOp::Multiply: main.multiply(op.a, op.b); break;
            case BufferedOp::AllocateMultiplier: main.allocate_multiplier(op.some, op.l, op.r); break;
            case BufferedOp::Constrain: cons.push_back(&op.a); break;
            }
        }
        sets.push_back(std::move(cons));
    }

    if (sets.empty()) return;
    for (auto &c : sets) if (c.empty()) return;               // empty factor: empty product
    std::vector<size_t> idx(sets.size(), 0);
    for (;;) {
        LinearCombination prod = *sets[0][idx[0]];
        for (size_t k = 1; k < sets.size(); k++) { MulVars m = main.multiply(prod, *sets[k][idx[k]]); prod = LinearCombination(m.o); }
        main.constrain(prod);
        size_t k = sets.size();
        while (k > 0) { k--; if (++idx[k] < sets[k].size()) break; idx[k] = 0; if (k == 0) return; }
    }
}

}  // namespace bpg
