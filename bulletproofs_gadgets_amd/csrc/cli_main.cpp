// bpg_prover / bpg_verifier: native file drivers for the reference's CLI formats, written against the C ABI only
// (include/bpg.h; no library internals).  They restate reference src/bin/prover.rs:47-100 and src/bin/verifier.rs:46-101:
//
//     bpg_prover   NAME    reads NAME.gadgets / NAME.inst / NAME.wtns, writes NAME.coms / NAME.proof, prints #constraints
//     bpg_verifier NAME    reads NAME.gadgets / NAME.inst / NAME.coms / NAME.proof, prints true|false, exit code 0|1
//
//     bpg_prover | bpg_verifier --batch FILE [--gpus N] [--workers W]
//                          FILE lists one NAME per line - a batch of independent proofs (the reference's own batch is its CI workflow, prover then
//                          verifier over twelve stems: .github/workflows/integration_tests.yml:19-58).  One process per GPU (the command starts them
//                          itself, before it touches the GPU); rank r takes stems r, r + N, ..., deals them to W host threads (default 4) with an
//                          engine context each - a proof is 0.5 s of host work (parsing, assembly, blinding chain) and 30 ms of GPU - writes their
//                          files and reports to the parent, which prints one summary line per stem in file order.  Same files as N = 1, as W = 1
//                          and as one run per stem.
//
// One executable, dispatched on argv[0] (or on a first argument "prover" / "verifier").  Grammar: the seven gadget lines of
// src/lalrpop/gadget_grammar.lalrpop:6-85 plus OR [ { .. } { .. } ] blocks (prover.rs:202-238, verifier.rs:162-186).
// Blinding factors: 64 bytes of /dev/urandom reduced mod l per factor (the reference uses thread_rng()); with BPG_CLI_SEED set they
// come from SHAKE256(seed || counter) so that two runs - and the Python driver bulletproofs_gadgets_amd/cli.py - produce the same
// files.  BPG_CLI_RNG_SEED (64 hex digits) fixes the 32 bytes that replace upstream's thread_rng() inside prove().
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include <execinfo.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
#include "../../include/bpg.h"

namespace {
// OR blocks may nest (reference src/bin/prover.rs:219-234 recurses on them too); a .gadgets file is untrusted input, so the recursion is bounded:
// the reference's own stems nest one level deep
constexpr unsigned MAX_OR_NESTING = 64;

typedef std::vector<uint8_t> Bytes;

[[noreturn]] void fail(const std::string &what) { throw std::runtime_error(what); }
void chk(bpg_status s, const char *what) {
    if (s != BPG_OK) fail(std::string(what) + ": " + (bpg_last_error() ? bpg_last_error() : "error"));
}

// ---------------------------------------------------------------------------------------------- small helpers
Bytes from_hex(const std::string &s) {                   // hex::decode of the reference: even length, hex digits only, else it panics
    if (s.size() % 2) fail("odd number of hex digits");
    Bytes out(s.size() / 2);
    for (size_t i = 0; i < out.size(); i++) {
        if (!isxdigit((unsigned char)s[2 * i]) || !isxdigit((unsigned char)s[2 * i + 1])) fail("invalid hex digit");
        out[i] = (uint8_t)std::stoul(s.substr(2 * i, 2), nullptr, 16);
    }
    return out;
}
std::string to_hex(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
// "X12 = 0xabcd" lines of .inst / .wtns / .coms (assignment_parser.rs:117-150), file order kept
std::vector<std::pair<std::string, Bytes>> read_vars(const std::string &path) {
    std::ifstream f(path);
    if (!f) fail("cannot open " + path);
    std::vector<std::pair<std::string, Bytes>> out;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::string name, eq, val;
        if (!(is >> name)) continue;
        if (!(is >> eq >> val) || eq != "=" || val.size() < 3 || val[0] != '0' || (val[1] != 'x' && val[1] != 'X')) fail("cannot parse '" + line + "' in " + path);
        out.emplace_back(name, from_hex(val.substr(2)));
    }
    return out;
}
uint64_t round_pow2(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }

// SHAKE256 for the reproducible blinding stream (FIPS 202; 64 output bytes fit one squeeze)
void keccak_f(uint64_t s[25]) {
    static const uint64_t RC[24] = {0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL, 0x8000000080008081ULL,
        0x8000000000008009ULL, 0x8aULL, 0x88ULL, 0x80008009ULL, 0x8000000aULL, 0x8000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x800aULL, 0x800000008000000aULL, 0x8000000080008081ULL,
        0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
    static const int rot[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int pil[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
        uint64_t c[5];
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) { uint64_t d = c[(x + 4) % 5] ^ ((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)); for (int y = 0; y < 25; y += 5) s[y + x] ^= d; }
        uint64_t t = s[1];
        for (int i = 0; i < 24; i++) { int j = pil[i]; uint64_t b = s[j]; s[j] = (t << rot[i]) | (t >> (64 - rot[i])); t = b; }
        for (int y = 0; y < 25; y += 5) { uint64_t a[5]; for (int x = 0; x < 5; x++) a[x] = s[y + x]; for (int x = 0; x < 5; x++) s[y + x] = a[x] ^ (~a[(x + 1) % 5] & a[(x + 2) % 5]); }
        s[0] ^= RC[r];
    }
}
void shake256_64(const Bytes &in, uint8_t out[64]) {
    uint64_t st[25] = {0}; uint8_t *b = reinterpret_cast<uint8_t *>(st); size_t pos = 0;
    for (uint8_t x : in) { b[pos++] ^= x; if (pos == 136) { keccak_f(st); pos = 0; } }
    b[pos] ^= 0x1f; b[135] ^= 0x80; keccak_f(st);
    std::memcpy(out, b, 64);
}

struct Blindings {                   // Scalar::random(&mut thread_rng()) stand-in (gadget.rs:31, commitments.rs:27,39)
    Bytes seed; bool seeded = false; uint64_t ctr = 0;
    Blindings() { if (const char *e = std::getenv("BPG_CLI_SEED")) { seed.assign(e, e + std::strlen(e)); seeded = true; } }
    Bytes next() {
        uint8_t wide[64];
        if (seeded) { Bytes in = seed; for (int i = 0; i < 8; i++) in.push_back((uint8_t)(ctr >> (8 * i))); ctr++; shake256_64(in, wide); }
        else { std::ifstream r("/dev/urandom", std::ios::binary); if (!r.read(reinterpret_cast<char *>(wide), 64)) fail("cannot read /dev/urandom"); }
        Bytes out(32);
        chk(bpg_scalar_op(5, wide, nullptr, out.data()), "from_wide");
        return out;
    }
    Bytes take(size_t k) { Bytes out; for (size_t i = 0; i < k; i++) { Bytes b = next(); out.insert(out.end(), b.begin(), b.end()); } return out; }
};

std::vector<Bytes> be_to_scalars(const Bytes &be) {          // conversions::be_to_scalars
    uint64_t n = be.size() / 32 + 2; Bytes buf(32 * n);
    chk(bpg_be_to_scalars(be.data(), be.size(), buf.data(), &n), "be_to_scalars");
    std::vector<Bytes> out;
    for (uint64_t i = 0; i < n; i++) out.emplace_back(buf.begin() + 32 * i, buf.begin() + 32 * i + 32);
    return out;
}
Bytes be_to_scalar(const Bytes &be) {                        // conversions::be_to_scalar: at most 32 bytes
    if (be.size() > 32) fail("value is longer than 32 bytes");
    Bytes le(32, 0);
    for (size_t i = 0; i < be.size(); i++) le[i] = be[be.size() - 1 - i];
    le[31] &= 0x7f;
    return le;
}
Bytes mimc_hash(const Bytes &pre) { Bytes o(32); chk(bpg_mimc_hash(pre.data(), pre.size(), o.data()), "mimc_hash"); return o; }
Bytes scalar_to_be(const Bytes &le) { return Bytes(le.rbegin(), le.rend()); }

// LinearCombination with one term: a variable (coefficient 1) or a constant
struct Lc {
    bpg_term t;
    static Lc var(uint32_t v) { Lc l; l.t.var = v; std::memset(l.t.coeff, 0, 32); l.t.coeff[0] = 1; return l; }
    static Lc constant(const Bytes &s) { Lc l; l.t.var = BPG_VAR_ONE << 29; std::memcpy(l.t.coeff, s.data(), 32); return l; }
};
std::vector<bpg_lc> views(const std::vector<Lc> &v) { std::vector<bpg_lc> o; for (const Lc &l : v) o.push_back(bpg_lc{&l.t, 1}); if (o.empty()) o.push_back(bpg_lc{nullptr, 0}); return o; }

// the dyn ConstraintSystem a gadget is proven / verified on: the real prover or verifier, or the recording buffer of an OR block
struct Cs { bpg_prover *p = nullptr; bpg_verifier *v = nullptr; bpg_buffer *b = nullptr; };

struct Gadget {
    bpg_gadget *h = nullptr;
    ~Gadget() { if (h) bpg_gadget_free(h); }
    // Gadget::setup (gadget.rs:18-38): commitments go to the real prover
    void setup(bpg_prover *p, const std::vector<Bytes> &wit, const Bytes &blind, std::vector<Bytes> &coms, std::vector<Bytes> &dsc, std::vector<uint32_t> &dvars) {
        Bytes w; for (const Bytes &x : wit) w.insert(w.end(), x.begin(), x.end());
        uint64_t cap = std::max<uint64_t>(8, std::max<uint64_t>(blind.size() / 32, 2 * wit.size() + 2)), n = cap;
        Bytes c(32 * cap), d(32 * cap); std::vector<uint32_t> v(cap);
        chk(bpg_gadget_setup(h, p, w.data(), wit.size(), blind.data(), blind.size() / 32, c.data(), d.data(), v.data(), &n), "Gadget::setup");
        for (uint64_t i = 0; i < n; i++) { coms.emplace_back(c.begin() + 32 * i, c.begin() + 32 * i + 32); dsc.emplace_back(d.begin() + 32 * i, d.begin() + 32 * i + 32); dvars.push_back(v[i]); }
    }
    void prove(const Cs &cs, const std::vector<uint32_t> &vars, const std::vector<Bytes> &dsc, const std::vector<uint32_t> &dvars) {
        Bytes d; for (const Bytes &x : dsc) d.insert(d.end(), x.begin(), x.end());
        uint32_t z = 0;
        if (cs.b) chk(bpg_gadget_prove_buffered(h, cs.b, vars.empty() ? &z : vars.data(), vars.size(), d.data(), dvars.empty() ? &z : dvars.data(), dvars.size()), "Gadget::prove");
        else chk(bpg_gadget_prove(h, cs.p, vars.empty() ? &z : vars.data(), vars.size(), d.data(), dvars.empty() ? &z : dvars.data(), dvars.size()), "Gadget::prove");
    }
    void verify(const Cs &cs, const std::vector<uint32_t> &vars, const std::vector<uint32_t> &dvars) {
        uint32_t z = 0;
        if (cs.b) chk(bpg_gadget_verify_buffered(h, cs.b, vars.empty() ? &z : vars.data(), vars.size(), dvars.empty() ? &z : dvars.data(), dvars.size()), "Gadget::verify");
        else chk(bpg_gadget_verify(h, cs.v, vars.empty() ? &z : vars.data(), vars.size(), dvars.empty() ? &z : dvars.data(), dvars.size()), "Gadget::verify");
    }
};
void new_bounds(Gadget &g, const Bytes &lo, const Bytes &hi) { chk(bpg_bounds_check_new(lo.data(), lo.size(), hi.data(), hi.size(), &g.h), "BoundsCheck::new"); }
void new_mimc(Gadget &g, const Lc &image) { bpg_lc v{&image.t, 1}; chk(bpg_mimc_hash256_new(&v, &g.h), "MimcHash256::new"); }
void new_merkle(Gadget &g, const Lc &root, const std::vector<Lc> &inst, const std::vector<Lc> &wit, const std::string &pattern) {
    bpg_lc r{&root.t, 1}; std::vector<bpg_lc> iv = views(inst), wv = views(wit);
    chk(bpg_merkle_tree256_new(&r, iv.data(), inst.size(), wv.data(), wit.size(), pattern.c_str(), &g.h), "MerkleTree256::new");
}
Bytes join(const std::vector<Bytes> &v) { Bytes o; for (const Bytes &x : v) o.insert(o.end(), x.begin(), x.end()); return o; }

// tree syntax of gadget_grammar.lalrpop:54-79 -> instance names, witness names, pattern string (left to right)
struct Tree { std::vector<std::string> inst, wit; std::string pattern; };
Tree parse_tree(const std::string &text) {
    std::vector<std::string> toks;
    for (size_t i = 0; i < text.size();) {
        char c = text[i];
        if (c == '(' || c == ')') { toks.emplace_back(1, c); i++; }
        else if (c == 'W' || c == 'I') { size_t j = i + 1; while (j < text.size() && isdigit((unsigned char)text[j])) j++; toks.push_back(text.substr(i, j - i)); i = j; }
        else i++;
    }
    Tree t; size_t pos = 0;
    struct Rec { std::vector<std::string> &toks; size_t &pos; Tree &t;
        std::string node(unsigned depth = 0) {
            if (depth > 64) fail("malformed tree: nested deeper than 64 levels");      // the recursion is bounded by the INPUT otherwise (fuzzer: stack overflow on ten thousand brackets)
            if (pos >= toks.size()) fail("malformed tree");
            std::string k = toks[pos++];
            if (k == "(") { std::string l = node(depth + 1), r = node(depth + 1); if (pos >= toks.size() || toks[pos] != ")") fail("malformed tree"); pos++; return "(" + l + " " + r + ")"; }
            if (k[0] == 'W') { t.wit.push_back(k); return "W"; }
            if (k[0] == 'I') { t.inst.push_back(k); return "I"; }
            fail("malformed tree");
        } } rec{toks, pos, t};
    t.pattern = rec.node();
    if (pos != toks.size() || t.pattern[0] != '(') fail("malformed tree");
    return t;
}
std::vector<std::string> split(const std::string &line) { std::istringstream is(line); std::vector<std::string> p; std::string w; while (is >> w) p.push_back(w); return p; }
std::string after_two(const std::string &line) {     // the rest of the line after the first two words (MERKLE root tree..)
    size_t i = 0; for (int k = 0; k < 2; k++) { while (i < line.size() && isspace((unsigned char)line[i])) i++; while (i < line.size() && !isspace((unsigned char)line[i])) i++; }
    return line.substr(i);
}
std::vector<std::string> read_lines(const std::string &path) {
    std::ifstream f(path); if (!f) fail("cannot open " + path);
    std::vector<std::string> out; std::string l; while (std::getline(f, l)) out.push_back(l); return out;
}

struct BufferGuard { bpg_buffer *b = nullptr; ~BufferGuard() { if (b) bpg_buffer_free(b); } };

// ================================================================================================ prover (prover.rs:47-100)
struct Witness { std::vector<Bytes> scalars, coms; std::vector<uint32_t> vars; Bytes data; };

// Two passes over the .gadgets file (default; BPG_CLI_TWO_PASS=0 gives the reference's single pass).  Every commitment a gadget line makes -
// Gadget::setup's derived witnesses, hash_witness's image - depends on witness and instance bytes only, never on constraints.  Pass 1 makes
// them all, in file order (same transcript, same blinding draws, same .coms lines as the single pass); then the transcript is final and the
// 2n serial TranscriptRng draws of prove() can start on the context's chain worker (bpg_prover_start_blinding) WHILE pass 2 assembles the
// constraints, replaying the cached commitments.  Same .coms and .proof bytes; the 0.3 s chain of a 2^20 circuit hides under the assembly.
struct Setup { std::vector<Bytes> coms, dsc; std::vector<uint32_t> dvars; };
struct ProverRun {
    std::string name; bpg_ctx *ctx = nullptr; bpg_transcript *tr = nullptr; bpg_prover *p = nullptr;
    std::map<std::string, Bytes> instance; std::map<std::string, Witness> witness; std::vector<std::string> lines;
    std::vector<std::pair<std::string, uint32_t>> coms_lines;      // ("C3-0 = 0x", committed variable): the bytes are read after the one batched commitment launch
    Blindings rnd;
    bool own_ctx = true, quiet = false;              // batch mode: the rank's context is shared by its stems, results go to the summary
    bool assemble_only = false;                      // tests/hostcheck (sanitizer / fuzz builds): parse and assemble without a device, stop before the proof
    uint64_t out_constraints = 0, out_proof_len = 0;
    int pass = 0;                                    // 0: single pass (commit and assemble line by line); 1: commitments only; 2: assembly only
    std::vector<Setup> cache; size_t cache_pos = 0;  // the commitments of pass 1 in the order pass 2 asks for them
    uint64_t est_multipliers = 0; bool saw_or = false;

    // Gadget::setup through the cache
    void setup(Gadget &g, const std::vector<Bytes> &wit, size_t nblind, std::vector<Bytes> &coms, std::vector<Bytes> &dsc, std::vector<uint32_t> &dvars) {
        if (pass == 2) { if (cache_pos >= cache.size()) fail("two-pass driver: commitment cache exhausted"); const Setup &c = cache[cache_pos++]; coms = c.coms; dsc = c.dsc; dvars = c.dvars; return; }
        g.setup(p, wit, rnd.take(nblind), coms, dsc, dvars);
        if (pass == 1) cache.push_back(Setup{coms, dsc, dvars});
    }
    void prove(Gadget &g, const Cs &cs, const std::vector<uint32_t> &vars, const std::vector<Bytes> &dsc, const std::vector<uint32_t> &dvars) { if (pass != 1) g.prove(cs, vars, dsc, dvars); }

    const Witness &single(const std::string &w) { const Witness &x = witness.at(w); if (x.scalars.size() != 1) fail("witness var " + w + " is longer than 32 bytes"); return x; }
    Lc lc_of(const std::string &tok) { if (tok[0] == 'W') return Lc::var(single(tok).vars[0]); const Bytes &d = instance.at(tok); if (d.size() > 32) fail("instance var " + tok + " is longer than 32 bytes"); return Lc::constant(be_to_scalar(d)); }
    void derived_lines(const std::vector<uint32_t> &vars, size_t index, int sub) { if (pass == 2) return; for (size_t k = 0; k < vars.size(); k++) coms_lines.emplace_back("D" + std::to_string(index) + "-" + std::to_string(sub) + "-" + std::to_string(k) + " = 0x", vars[k]); }
    uint64_t cs_next_multiplier(const Cs &cs) { return cs.b ? bpg_buffer_next_multiplier(cs.b) : bpg_prover_num_multiplications(cs.p); }

    // hash_witness (prover.rs:160-190): commit to the MiMC image of a witness and prove the preimage relation -> (image scalar, image var)
    std::pair<Bytes, uint32_t> hash_witness(const std::string &wn, size_t index, int sub, const Cs &cs) {
        const Witness &w = witness.at(wn);
        Bytes image = mimc_hash(w.data), com(32); uint32_t var = 0;
        if (pass == 2) { if (cache_pos >= cache.size()) fail("two-pass driver: commitment cache exhausted"); const Setup &c = cache[cache_pos++]; com = c.coms.at(0); var = c.dvars.at(0); }
        else {
            Bytes blind = rnd.next(), image_be = scalar_to_be(image), image_sc = be_to_scalar(image_be);
            chk(bpg_prover_commit(p, image_sc.data(), blind.data(), com.data(), &var), "Prover::commit");
            if (pass == 1) cache.push_back(Setup{{com}, {}, {var}});
        }
        est_multipliers += 972 * (w.scalars.size() + 1);
        Gadget g; new_mimc(g, Lc::var(var));
        std::vector<Bytes> dcoms, dsc; std::vector<uint32_t> dvars;
        setup(g, w.scalars, 2, dcoms, dsc, dvars);
        prove(g, cs, w.vars, dsc, dvars);
        std::vector<uint32_t> all{var}; all.insert(all.end(), dvars.begin(), dvars.end());
        derived_lines(all, index, sub);
        return {image, var};
    }

    void do_gadget(const std::string &line, size_t index, const Cs &cs) {
        std::vector<std::string> parts = split(line);
        const std::string &op = parts[0];
        std::vector<Bytes> dcoms, dsc; std::vector<uint32_t> dvars;
        Gadget g;
        if (op == "BOUND") {                                              // prover.rs:253-276
            const Witness &w = single(parts.at(1));
            new_bounds(g, instance.at(parts.at(2)), instance.at(parts.at(3)));
            est_multipliers += 16 * instance.at(parts.at(3)).size();
            setup(g, w.scalars, 2, dcoms, dsc, dvars); prove(g, cs, w.vars, dsc, dvars); derived_lines(dvars, index, 0);
        } else if (op == "HASH") {                                        // prover.rs:278-305
            new_mimc(g, lc_of(parts.at(1)));
            const Witness &w = witness.at(parts.at(2));
            est_multipliers += 972 * (w.scalars.size() + 1);
            setup(g, w.scalars, 2, dcoms, dsc, dvars); prove(g, cs, w.vars, dsc, dvars); derived_lines(dvars, index, 0);
        } else if (op == "MERKLE") {                                      // prover.rs:307-339
            Lc root = lc_of(parts.at(1));
            Tree t = parse_tree(after_two(line));
            std::vector<Lc> il, wl;
            for (const std::string &i : t.inst) il.push_back(Lc::constant(mimc_hash(instance.at(i))));
            int sub = 0; for (const std::string &wn : t.wit) wl.push_back(Lc::var(hash_witness(wn, index, sub++, cs).second));
            new_merkle(g, root, il, wl, t.pattern);
            for (char c : t.pattern) if (c == '(') est_multipliers += 1944;
            prove(g, cs, {}, {}, {});
        } else if (op == "EQUALS") {                                      // prover.rs:340-358 (grammar: W I | I W | W W)
            std::string left = parts.at(1), right = parts.at(2); if (left[0] != 'W') std::swap(left, right);
            std::vector<Lc> rl;
            if (right[0] == 'W') for (uint32_t v : witness.at(right).vars) rl.push_back(Lc::var(v)); else for (const Bytes &s : be_to_scalars(instance.at(right))) rl.push_back(Lc::constant(s));
            std::vector<bpg_lc> rv = views(rl);
            chk(bpg_equality_new(rv.data(), rl.size(), &g.h), "Equality::new");
            prove(g, cs, witness.at(left).vars, {}, {});
        } else if (op == "LESS_THAN") {                                   // prover.rs:360-382
            const Witness &l = single(parts.at(1)), &r = single(parts.at(2));
            Lc ll = Lc::var(l.vars[0]), rl = Lc::var(r.vars[0]); bpg_lc lv{&ll.t, 1}, rv{&rl.t, 1};
            chk(bpg_less_than_new(&lv, l.scalars[0].data(), &rv, r.scalars[0].data(), &g.h), "LessThan::new");
            est_multipliers += 379;
            setup(g, {}, 2, dcoms, dsc, dvars); prove(g, cs, {}, dsc, dvars); derived_lines(dvars, index, 0);
        } else if (op == "UNEQUAL") {                                     // prover.rs:384-418
            std::string left = parts.at(1), right = parts.at(2); if (left[0] != 'W') std::swap(left, right);
            const Witness &lw = witness.at(left);
            std::vector<Bytes> rs; std::vector<Lc> rl;
            if (right[0] == 'W') { rs = witness.at(right).scalars; for (uint32_t v : witness.at(right).vars) rl.push_back(Lc::var(v)); }
            else { rs = be_to_scalars(instance.at(right)); for (const Bytes &s : rs) rl.push_back(Lc::constant(s)); }
            std::vector<bpg_lc> rv = views(rl); Bytes ra = join(rs);
            chk(bpg_inequality_new(rv.data(), rl.size(), ra.data(), &g.h), "Inequality::new");
            est_multipliers += 2 * lw.scalars.size() + 1;
            setup(g, lw.scalars, 2 * lw.scalars.size() + 1, dcoms, dsc, dvars); prove(g, cs, lw.vars, dsc, dvars); derived_lines(dvars, index, 0);
        } else if (op == "SET_MEMBER") {                                  // prover.rs:420-532
            const std::string &member = parts.at(1); std::vector<std::string> elems(parts.begin() + 2, parts.end());
            std::vector<Bytes> m_scalars; std::vector<Lc> m_lcs;
            if (member[0] == 'W') { m_scalars = witness.at(member).scalars; for (uint32_t v : witness.at(member).vars) m_lcs.push_back(Lc::var(v)); }
            else { m_scalars = be_to_scalars(instance.at(member)); for (const Bytes &s : m_scalars) m_lcs.push_back(Lc::constant(s)); }
            Bytes m_scalar = m_scalars.at(0); Lc m_lc = m_lcs.at(0);
            bool hashing = m_scalars.size() > 1;
            std::vector<uint32_t> w_vars; std::vector<Bytes> w_scalars, i_scalars; std::vector<Lc> i_lcs;
            if (!hashing) for (const std::string &e : elems) {
                if (e[0] == 'W') { const Witness &w = witness.at(e); if (w.vars.size() == 1) { w_scalars.push_back(w.scalars[0]); w_vars.push_back(w.vars[0]); } else hashing = true; }
                else { std::vector<Bytes> sc = be_to_scalars(instance.at(e)); if (sc.size() == 1) { i_scalars.push_back(sc[0]); i_lcs.push_back(Lc::constant(sc[0])); } else hashing = true; }
            }
            if (hashing) {                                                // elements longer than one scalar: compare MiMC images
                int sub = 1;
                if (member[0] == 'W') { auto h = hash_witness(member, index, sub++, cs); m_scalar = h.first; m_lc = Lc::var(h.second); }
                else { m_scalar = mimc_hash(instance.at(member)); m_lc = Lc::constant(m_scalar); }
                w_vars.clear(); w_scalars.clear(); i_scalars.clear(); i_lcs.clear();
                for (const std::string &e : elems) {
                    if (e[0] == 'W') { auto h = hash_witness(e, index, sub++, cs); w_vars.push_back(h.second); w_scalars.push_back(h.first); }
                    else { Bytes h = mimc_hash(instance.at(e)); i_lcs.push_back(Lc::constant(h)); i_scalars.push_back(h); }
                }
            }
            bpg_lc mv{&m_lc.t, 1}; std::vector<bpg_lc> iv = views(i_lcs); Bytes ia = join(i_scalars);
            chk(bpg_set_membership_new(&mv, m_scalar.data(), iv.data(), i_lcs.size(), ia.data(), &g.h), "SetMembership::new");
            est_multipliers += 2 * (w_scalars.size() + i_scalars.size());
            setup(g, w_scalars, w_scalars.size() + i_scalars.size(), dcoms, dsc, dvars); prove(g, cs, w_vars, dsc, dvars); derived_lines(dvars, index, 0);
        } else fail("unknown gadget line: '" + line + "'");
    }

    // lines from i on; closing = 0 at top level, ']' inside an OR block (prover.rs:75-84 and :219-234)
    size_t run_block(size_t i, const Cs &cs, char closing, unsigned depth = 0) {
        if (depth > MAX_OR_NESTING) fail("OR blocks nested deeper than " + std::to_string(MAX_OR_NESTING));      // the file is untrusted input: bounded recursion
        while (i < lines.size()) {
            const std::string &line = lines[i]; size_t index = i++;
            std::vector<std::string> parts = split(line);
            if (parts.empty()) continue;
            const std::string &op = parts[0];
            if (closing && op.size() == 1 && op[0] == closing) return i;
            if (pass == 1) {                                  // commitments only: clauses of OR blocks are walked in file order, nothing is recorded
                if (op == "OR") { saw_or = true; i = run_block(i, cs, ']', depth + 1); }
                else if (op == "}" || op == "[" || op == "{") { }
                else do_gadget(line, index, cs);
                continue;
            }
            if (op == "}") { if (!cs.b) fail("'}' outside an OR block"); chk(bpg_buffer_rewind(cs.b), "rewind"); }
            else if (op == "OR") {
                BufferGuard child; chk(bpg_buffer_new(cs_next_multiplier(cs), 1, &child.b), "ProverBuffer::new");
                Cs inner; inner.b = child.b;
                i = run_block(i, inner, ']', depth + 1);
                if (cs.b) chk(bpg_or_buffer(cs.b, child.b), "or"); else chk(bpg_or_prover(cs.p, child.b), "or");
            } else if (op == "[" || op == "{") { /* block openers */ }
            else do_gadget(line, index, cs);
        }
        if (closing) fail("unexpected end of input");
        return i;
    }

    int run() {
        const bool timing = std::getenv("BPG_CLI_TIMING") != nullptr;         // phase times on stderr
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        auto lap = [&](const char *what) { if (timing) { double t1 = now(); std::fprintf(stderr, "  %-28s %8.2f ms\n", what, t1 - t0); t0 = t1; } };
        if (!ctx && !assemble_only) chk(bpg_ctx_create(0, &ctx), "bpg_ctx_create");
        lap("context (HIP init, bases)");
        chk(bpg_transcript_new(reinterpret_cast<const uint8_t *>(name.data()), name.size(), &tr), "Transcript::new");
        chk(bpg_prover_new(ctx, tr, &p), "Prover::new");
        if (assemble_only) chk(bpg_test_prover_stub_commitments(p), "stub commitments");        // tests/hostcheck: no device, hash bytes for commitments
        // every Pedersen commitment of the run - witness variables, hash_witness images, Gadget::setup's derived values: about a thousand for a
        // 256-leaf tree - in ONE kernel launch when the last one has been registered (bpg.h bpg_prover_defer_commitments): same transcript,
        // same .coms bytes as one launch per commitment
        chk(bpg_prover_defer_commitments(p, 1), "defer_commitments");
        for (auto &kv : read_vars(name + ".inst")) instance[kv.first] = kv.second;
        for (auto &kv : read_vars(name + ".wtns")) {                      // assignment_parser.rs:152-169
            Witness w; w.data = kv.second; w.scalars = be_to_scalars(kv.second);
            const size_t k = w.scalars.size();
            Bytes v = join(w.scalars), blind = rnd.take(std::max<size_t>((kv.second.size() + 31) / 32, 1)), coms(32 * k); w.vars.resize(k);
            chk(bpg_prover_commit_many(p, k, v.data(), blind.data(), coms.data(), w.vars.data()), "Prover::commit");
            for (size_t j = 0; j < k; j++) coms_lines.emplace_back("C" + kv.first.substr(1) + "-" + std::to_string(j) + " = 0x", w.vars[j]);
            witness[kv.first] = w;
        }
        lap("witness values");
        lines = read_lines(name + ".gadgets");
        Bytes rng_seed(32);
        if (const char *e = std::getenv("BPG_CLI_RNG_SEED")) { rng_seed = from_hex(e); rng_seed.resize(32); }
        else { std::ifstream r("/dev/urandom", std::ios::binary); if (!r.read(reinterpret_cast<char *>(rng_seed.data()), 32)) fail("cannot read /dev/urandom"); }
        Cs top; top.p = p;
        const char *tp = std::getenv("BPG_CLI_TWO_PASS");
        if (tp && std::atoi(tp) == 0) { pass = 0; run_block(0, top, 0); }
        else {
            pass = 1; run_block(0, top, 0);
            lap("gadget values (pass 1)");
            chk(bpg_prover_flush_commitments(p), "flush_commitments");
            lap("all commitments (one launch)");
            // the transcript is final: start the chain of prove() now.  The estimate only sizes the pinned buffer; a stream that turns out too
            // short (OR blocks add product multipliers) is simply not used
            uint64_t est = est_multipliers + 64; if (saw_or) est = 2 * est + 65536;
            if (!assemble_only) chk(bpg_prover_start_blinding(p, rng_seed.data(), est), "start_blinding");
            // the generators (0.11 s to derive at 2^20, once per device) are made while pass 2 assembles: nothing in pass 2 touches the context.
            // The capacity is the estimate's; prove() below asks for the real one, which grows the tables should the estimate have been short
            struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } early;
            if (!saw_or && est_multipliers && !assemble_only) { bpg_ctx *c = ctx; const uint64_t cap_est = round_pow2(est_multipliers); early.t = std::thread([c, cap_est] { (void)bpg_gens_ensure(c, cap_est); }); }
            pass = 2; run_block(0, top, 0);
            if (early.t.joinable()) early.t.join();
            if (cache_pos != cache.size()) fail("two-pass driver: pass 2 did not use every commitment of pass 1");
        }
        lap("gadget assembly");
        out_constraints = bpg_prover_num_constraints(p);
        if (!quiet) { std::printf("%llu\n", (unsigned long long)out_constraints); std::fflush(stdout); }          // prover.rs:89
        const uint64_t n = bpg_prover_num_multiplications(p), cap = round_pow2(n);
        if (assemble_only) { out_proof_len = 0; return 0; }
        chk(bpg_gens_ensure(ctx, cap), "BulletproofGens::new");
        lap("generators");
        uint64_t plen = bpg_proof_size(n, 0); Bytes proof(plen);
        chk(bpg_prover_prove(p, cap, rng_seed.data(), 0, proof.data(), &plen, nullptr), "Prover::prove");
        lap("prove (upload + proof)");
        {
            std::ofstream f(name + ".coms"); uint8_t com[32];
            for (const auto &l : coms_lines) { chk(bpg_prover_commitment(p, l.second & 0x1fffffffu, com), "commitment"); f << l.first << to_hex(com, 32) << "\n"; }
        }
        { std::ofstream f(name + ".proof", std::ios::binary); f.write(reinterpret_cast<const char *>(proof.data()), (std::streamsize)plen); }
        out_proof_len = plen;
        return 0;
    }
    ~ProverRun() { if (p) bpg_prover_free(p); if (tr) bpg_transcript_free(tr); if (ctx && own_ctx) bpg_ctx_destroy(ctx); }
};

// ================================================================================================ verifier (verifier.rs:46-101)
struct VerifierRun {
    std::string name; bpg_ctx *ctx = nullptr; bpg_transcript *tr = nullptr; bpg_verifier *v = nullptr;
    bool own_ctx = true, quiet = false;
    bool assemble_only = false;                      // tests/hostcheck: replay without a device, stop before verify()
    std::map<std::string, Bytes> instance; std::map<std::string, uint32_t> commitments; std::vector<std::string> lines;

    std::vector<uint32_t> all_commitments(const std::string &w) {
        std::vector<uint32_t> out;
        for (size_t k = 0;; k++) { auto it = commitments.find("C" + w.substr(1) + "-" + std::to_string(k)); if (it == commitments.end()) break; out.push_back(it->second); }
        if (out.empty()) fail("missing commitment C" + w.substr(1) + "-0");
        return out;
    }
    uint32_t com(const std::string &key) { auto it = commitments.find(key); if (it == commitments.end()) fail("missing commitment " + key); return it->second; }
    Lc lc_of(const std::string &tok) { if (tok[0] == 'W') return Lc::var(com("C" + tok.substr(1) + "-0")); return Lc::constant(be_to_scalar(instance.at(tok))); }
    std::vector<uint32_t> derived(size_t index, int sub, size_t upto) {
        std::vector<uint32_t> out;
        for (size_t k = 0; k < upto; k++) { auto it = commitments.find("D" + std::to_string(index) + "-" + std::to_string(sub) + "-" + std::to_string(k)); if (it == commitments.end()) break; out.push_back(it->second); }
        return out;
    }
    std::vector<uint32_t> derived_exact(size_t index, size_t count) { std::vector<uint32_t> out; for (size_t k = 0; k < count; k++) out.push_back(com("D" + std::to_string(index) + "-0-" + std::to_string(k))); return out; }
    uint64_t cs_next_multiplier(const Cs &cs) { return cs.b ? bpg_buffer_next_multiplier(cs.b) : bpg_verifier_num_vars(cs.v); }
    uint32_t hash_witness(const std::string &wn, size_t index, int sub, const Cs &cs) {     // verifier.rs:426-444
        std::vector<uint32_t> d = derived(index, sub, 3);
        if (d.empty()) fail("missing derived commitments of line " + std::to_string(index));
        Gadget g; new_mimc(g, Lc::var(d[0]));
        g.verify(cs, all_commitments(wn), std::vector<uint32_t>(d.begin() + 1, d.end()));
        return d[0];
    }
    void do_gadget(const std::string &line, size_t index, const Cs &cs) {
        std::vector<std::string> parts = split(line);
        const std::string &op = parts[0];
        Gadget g;
        if (op == "BOUND") { new_bounds(g, instance.at(parts.at(2)), instance.at(parts.at(3))); g.verify(cs, {com("C" + parts.at(1).substr(1) + "-0")}, derived(index, 0, 2)); }
        else if (op == "HASH") { new_mimc(g, lc_of(parts.at(1))); g.verify(cs, all_commitments(parts.at(2)), derived(index, 0, 2)); }
        else if (op == "MERKLE") {
            Lc root = lc_of(parts.at(1)); Tree t = parse_tree(after_two(line));
            std::vector<Lc> il, wl;
            for (const std::string &i : t.inst) il.push_back(Lc::constant(mimc_hash(instance.at(i))));
            int sub = 0; for (const std::string &wn : t.wit) wl.push_back(Lc::var(hash_witness(wn, index, sub++, cs)));
            new_merkle(g, root, il, wl, t.pattern); g.verify(cs, {}, {});
        } else if (op == "EQUALS") {
            std::string left = parts.at(1), right = parts.at(2); if (left[0] != 'W') std::swap(left, right);
            std::vector<Lc> rl;
            if (right[0] == 'W') for (uint32_t x : all_commitments(right)) rl.push_back(Lc::var(x)); else for (const Bytes &s : be_to_scalars(instance.at(right))) rl.push_back(Lc::constant(s));
            std::vector<bpg_lc> rv = views(rl);
            chk(bpg_equality_new(rv.data(), rl.size(), &g.h), "Equality::new");
            g.verify(cs, all_commitments(left), {});
        } else if (op == "LESS_THAN") {
            Lc ll = Lc::var(com("C" + parts.at(1).substr(1) + "-0")), rl = Lc::var(com("C" + parts.at(2).substr(1) + "-0")); bpg_lc lv{&ll.t, 1}, rv{&rl.t, 1};
            chk(bpg_less_than_new(&lv, nullptr, &rv, nullptr, &g.h), "LessThan::new");
            g.verify(cs, {}, derived(index, 0, 2));
        } else if (op == "UNEQUAL") {
            std::string left = parts.at(1), right = parts.at(2); if (left[0] != 'W') std::swap(left, right);
            std::vector<uint32_t> lv = all_commitments(left);
            std::vector<Lc> rl;
            if (right[0] == 'W') for (uint32_t x : all_commitments(right)) rl.push_back(Lc::var(x)); else for (const Bytes &s : be_to_scalars(instance.at(right))) rl.push_back(Lc::constant(s));
            std::vector<bpg_lc> rv = views(rl);
            chk(bpg_inequality_new(rv.data(), rl.size(), nullptr, &g.h), "Inequality::new");
            g.verify(cs, lv, derived_exact(index, 2 * lv.size() + 1));
        } else if (op == "SET_MEMBER") {
            const std::string &member = parts.at(1); std::vector<std::string> elems(parts.begin() + 2, parts.end());
            std::vector<Lc> m_lcs;
            if (member[0] == 'W') for (uint32_t x : all_commitments(member)) m_lcs.push_back(Lc::var(x)); else for (const Bytes &s : be_to_scalars(instance.at(member))) m_lcs.push_back(Lc::constant(s));
            Lc m_lc = m_lcs.at(0); bool hashing = m_lcs.size() > 1;
            std::vector<uint32_t> w_vars; std::vector<Lc> i_lcs;
            for (const std::string &e : elems) {
                if (e[0] == 'W') { std::vector<uint32_t> cw = all_commitments(e); if (cw.size() == 1) w_vars.push_back(cw[0]); else hashing = true; }
                else { std::vector<Bytes> sc = be_to_scalars(instance.at(e)); if (sc.size() == 1) i_lcs.push_back(Lc::constant(sc[0])); else hashing = true; }
            }
            std::vector<uint32_t> d = derived_exact(index, elems.size());
            if (hashing) {
                int sub = 1;
                if (member[0] == 'W') m_lc = Lc::var(hash_witness(member, index, sub++, cs)); else m_lc = Lc::constant(mimc_hash(instance.at(member)));
                w_vars.clear(); i_lcs.clear();
                for (const std::string &e : elems) { if (e[0] == 'W') w_vars.push_back(hash_witness(e, index, sub++, cs)); else i_lcs.push_back(Lc::constant(mimc_hash(instance.at(e)))); }
            }
            bpg_lc mv{&m_lc.t, 1}; std::vector<bpg_lc> iv = views(i_lcs);
            chk(bpg_set_membership_new(&mv, nullptr, iv.data(), i_lcs.size(), nullptr, &g.h), "SetMembership::new");
            g.verify(cs, w_vars, d);
        } else fail("unknown gadget line: '" + line + "'");
    }
    size_t run_block(size_t i, const Cs &cs, char closing, unsigned depth = 0) {
        if (depth > MAX_OR_NESTING) fail("OR blocks nested deeper than " + std::to_string(MAX_OR_NESTING));
        while (i < lines.size()) {
            const std::string &line = lines[i]; size_t index = i++;
            std::vector<std::string> parts = split(line);
            if (parts.empty()) continue;
            const std::string &op = parts[0];
            if (closing && op.size() == 1 && op[0] == closing) return i;
            if (op == "}") { if (!cs.b) fail("'}' outside an OR block"); chk(bpg_buffer_rewind(cs.b), "rewind"); }
            else if (op == "OR") {                                        // verifier.rs:162-186
                BufferGuard child; chk(bpg_buffer_new(cs_next_multiplier(cs), 0, &child.b), "VerifierBuffer::new");
                Cs inner; inner.b = child.b;
                i = run_block(i, inner, ']', depth + 1);
                if (cs.b) chk(bpg_or_buffer(cs.b, child.b), "or"); else chk(bpg_or_verifier(cs.v, child.b), "or");
            } else if (op == "[" || op == "{") { }
            else do_gadget(line, index, cs);
        }
        if (closing) fail("unexpected end of input");
        return i;
    }
    int run() {
        chk(bpg_transcript_new(reinterpret_cast<const uint8_t *>(name.data()), name.size(), &tr), "Transcript::new");
        chk(bpg_verifier_new(tr, &v), "Verifier::new");
        std::ifstream pf(name + ".proof", std::ios::binary); if (!pf) fail("cannot open " + name + ".proof");
        Bytes proof((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
        for (auto &kv : read_vars(name + ".inst")) instance[kv.first] = kv.second;
        for (auto &kv : read_vars(name + ".coms")) {                      // parse_coms: every line, file order
            if (kv.second.size() != 32) fail("commitment " + kv.first + " is not 32 bytes");
            uint32_t var = 0; chk(bpg_verifier_commit(v, kv.second.data(), &var), "Verifier::commit"); commitments[kv.first] = var;
        }
        lines = read_lines(name + ".gadgets");
        Cs top; top.v = v;
        run_block(0, top, 0);
        if (assemble_only) return 0;
        if (!ctx) chk(bpg_ctx_create(0, &ctx), "bpg_ctx_create");
        const uint64_t cap = round_pow2(bpg_verifier_num_vars(v));
        chk(bpg_gens_ensure(ctx, cap), "BulletproofGens::new");
        Bytes seed(32); { std::ifstream r("/dev/urandom", std::ios::binary); r.read(reinterpret_cast<char *>(seed.data()), 32); }
        const bpg_status s = bpg_verifier_verify(v, ctx, cap, proof.data(), proof.size(), seed.data(), 0);
        // (flushed at once: the verdict must reach the caller whatever happens while the process winds down)
        if (s == BPG_OK) { if (!quiet) { std::puts("true"); std::fflush(stdout); } return 0; }                 // verifier.rs:91-100
        if (s == BPG_ERR_VERIFICATION || s == BPG_ERR_FORMAT) { if (!quiet) { std::puts("false"); std::fflush(stdout); } return 1; }
        fail(std::string("Verifier::verify: ") + (bpg_last_error() ? bpg_last_error() : "error"));
    }
    ~VerifierRun() { if (v) bpg_verifier_free(v); if (tr) bpg_transcript_free(tr); if (ctx && own_ctx) bpg_ctx_destroy(ctx); }
};

// ================================================================================================ batches
std::vector<std::string> read_batch(const std::string &path) {
    std::ifstream f(path);
    if (!f) fail("cannot open " + path);
    std::vector<std::string> out; std::string line;
    while (std::getline(f, line)) {
        size_t a = line.find_first_not_of(" \t\r"), b = line.find_last_not_of(" \t\r");
        if (a == std::string::npos || line[a] == '#') continue;
        out.push_back(line.substr(a, b - a + 1));
    }
    return out;
}
// one rank of a batch: stems rank, rank + world, ... on device `rank mod devices`, dealt to `workers` host threads with an engine context each (the
// contexts of a process share the generator tables of the device).  One proof keeps a host thread busy for its parsing, assembly and blinding chain
// (0.5 s at 2^20) and the GPU for 30 ms, so a few workers per GPU multiply the rate of a batch; every stem is still proved exactly as a lone
// `bpg_prover NAME` run would prove it (own transcript, own blinding stream).  A result line per stem on `out`:
// "<index>\t<constraints>\t<proof bytes>" (prover) or "<index>\t<true|false>" (verifier)
int run_batch_rank(const std::string &mode, const std::vector<std::string> &stems, uint32_t rank, uint32_t world, uint32_t workers, FILE *out) {
    const int32_t ndev = bpg_device_count();
    if (ndev <= 0) fail("no AMD GPU visible: the library has no CPU path");
    std::vector<size_t> mine;
    for (size_t i = rank; i < stems.size(); i += world) mine.push_back(i);
    if (workers < 1) workers = 1;
    if (workers > mine.size()) workers = (uint32_t)std::max<size_t>(mine.size(), 1);
    std::atomic<size_t> next(0);
    std::atomic<int> rc(0);
    std::mutex out_mu; std::string first_error;
    auto note_error = [&](const std::string &what) { std::lock_guard<std::mutex> lk(out_mu); if (first_error.empty()) first_error = what; rc.store(101); };
    auto work = [&]() {
        bpg_ctx *ctx = nullptr;
        try { chk(bpg_ctx_create((int32_t)(rank % (uint32_t)ndev), &ctx), "bpg_ctx_create"); }
        catch (const std::exception &e) { note_error(e.what()); return; }       // no context, no work: the stems go to the other workers
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= mine.size()) break;
            const size_t i = mine[k];
            char line[128]; std::string text;
            // ONE stem per try: a missing or garbled file ends that stem only, as it would end only its own run in the reference's CI
            // (one prover process per stem, .github/workflows/integration_tests.yml:19-58); the worker goes on with the next stem
            try {
                if (mode == "prover") {
                    ProverRun r; r.name = stems[i]; r.ctx = ctx; r.own_ctx = false; r.quiet = true;
                    r.run();
                    std::snprintf(line, sizeof line, "%zu\t%llu\t%llu\n", i, (unsigned long long)r.out_constraints, (unsigned long long)r.out_proof_len);
                } else {
                    VerifierRun r; r.name = stems[i]; r.ctx = ctx; r.own_ctx = false; r.quiet = true;
                    const int ok = r.run();
                    if (ok != 0) { int want = 0; rc.compare_exchange_strong(want, 1); }
                    std::snprintf(line, sizeof line, "%zu\t%s\n", i, ok == 0 ? "true" : "false");
                }
                text = line;
            } catch (const std::exception &e) {
                std::string msg = e.what();
                for (char &ch : msg) if (ch == '\n' || ch == '\t' || ch == '\r') ch = ' ';
                text = std::to_string(i) + "\tERROR\t" + msg + "\n";
                note_error(stems[i] + ": " + msg);
            }
            std::lock_guard<std::mutex> lk(out_mu);
            std::fputs(text.c_str(), out); std::fflush(out);
        }
        if (ctx) bpg_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (uint32_t w = 1; w < workers; w++) th.emplace_back(work);
    work();
    for (std::thread &t : th) t.join();
    if (!first_error.empty()) std::fprintf(stderr, "%s --batch (rank %u): %s\n", mode.c_str(), rank, first_error.c_str());
    return rc.load();
}
// a ROCm profiler library is preloaded into this process (it initialises the GPU before main(): see run_batch)
bool profiler_preloaded() {
    for (const char *var : {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB"}) {
        const char *v = std::getenv(var);
        if (!v) continue;
        if (std::string(var) == "ROCPROFILER_REGISTER_FORCE_LOAD") { if (*v && std::string(v) != "0") return true; continue; }
        const std::string s(v);
        if (s.find("rocprof") != std::string::npos || s.find("roctracer") != std::string::npos || s.find("rocprofiler") != std::string::npos) return true;
    }
    return false;
}
// the batch command: with one GPU it is the rank; with --gpus N it starts N ranks of this executable (nothing here has touched the GPU), reads their
// result lines from pipes and prints the summary in file order.  Exit code: 0, 1 when a proof was rejected, 101 when a rank failed (the reference panics).
int run_batch(const std::string &self_path, const std::string &mode, const std::string &file, uint32_t gpus, uint32_t workers) {
    const std::vector<std::string> stems = read_batch(file);
    std::vector<std::string> result(stems.size());
    int rc = 0;
    auto take = [&](const std::string &line) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) return;
        const size_t idx = std::stoul(line.substr(0, tab));
        if (idx < result.size()) result[idx] = line.substr(tab + 1);
    };
    if (gpus <= 1) {
        char *buf = nullptr; size_t len = 0;
        FILE *mem = open_memstream(&buf, &len);
        if (!mem) fail("open_memstream");
        try { rc = run_batch_rank(mode, stems, 0, 1, workers, mem); } catch (...) { std::fclose(mem); std::free(buf); throw; }
        std::fclose(mem);
        std::istringstream is(std::string(buf, len)); std::free(buf);
        for (std::string l; std::getline(is, l);) take(l);
    } else {
        struct Child { pid_t pid; int fd; };
        std::vector<Child> kids;
        for (uint32_t r = 0; r < gpus; r++) {
            int pfd[2];
            if (pipe(pfd) != 0) fail("pipe");
            const pid_t pid = fork();
            if (pid < 0) fail("fork");
            if (pid == 0) {                                  // the rank: a fresh image of this executable with its results on the pipe
                close(pfd[0]);
                dup2(pfd[1], 3); if (pfd[1] != 3) close(pfd[1]);
                const std::string rs = std::to_string(r), ws = std::to_string(gpus), ks = std::to_string(workers);
                const char *args[] = {self_path.c_str(), mode.c_str(), "--batch", file.c_str(), "--rank", rs.c_str(), "--world", ws.c_str(), "--workers", ks.c_str(), nullptr};
                execv(self_path.c_str(), const_cast<char *const *>(args));
                std::perror("execv"); _exit(127);
            }
            close(pfd[1]);
            kids.push_back(Child{pid, pfd[0]});
        }
        for (Child &k : kids) {
            std::string all; char buf[4096]; ssize_t n;
            while ((n = read(k.fd, buf, sizeof buf)) > 0) all.append(buf, (size_t)n);
            close(k.fd);
            int st = 0; waitpid(k.pid, &st, 0);
            const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 101;
            if (code == 1 && rc == 0) rc = 1; else if (code != 0 && code != 1) rc = 101;
            std::istringstream is(all);
            for (std::string l; std::getline(is, l);) take(l);
        }
    }
    for (size_t i = 0; i < stems.size(); i++) {
        if (result[i].empty()) { std::printf("%s: FAILED\n", stems[i].c_str()); rc = 101; continue; }
        if (result[i].compare(0, 6, "ERROR\t") == 0) { std::printf("%s: FAILED (%s)\n", stems[i].c_str(), result[i].substr(6).c_str()); rc = 101; continue; }
        if (mode == "prover") {
            const size_t tab = result[i].find('\t');
            std::printf("%s: %s constraints, %s-byte proof\n", stems[i].c_str(), result[i].substr(0, tab).c_str(), result[i].substr(tab + 1).c_str());
        } else std::printf("%s: %s\n", stems[i].c_str(), result[i].c_str());
    }
    return rc;
}

}  // namespace

#ifndef BPG_CLI_NO_MAIN
// a crash must not be silent: the frames go to stderr (async-signal-safe calls only), then the default action takes over
static void crash_report(int sig) {
    static const char msg[] = "bpg_prover/bpg_verifier: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    void *frames[48];
    backtrace_symbols_fd(frames, backtrace(frames, 48), 2);
    raise(sig);                                                               // SA_RESETHAND put the default action back: the process dies of the signal
}
int main(int argc, char **argv) {
    { void *warm[2]; (void)backtrace(warm, 2); }                              // loads the unwinder now: the handler must not allocate
    {   // on an alternate stack, so that a stack overflow reports too instead of dying silently inside its own handler
        static char altstack[1 << 16];
        stack_t ss; std::memset(&ss, 0, sizeof ss); ss.ss_sp = altstack; ss.ss_size = sizeof altstack;
        (void)sigaltstack(&ss, nullptr);
        struct sigaction sa; std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = crash_report; sa.sa_flags = SA_ONSTACK | SA_RESETHAND; sigemptyset(&sa.sa_mask);
        for (int sig : {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT}) (void)sigaction(sig, &sa, nullptr);
    }
    std::string self = argv[0]; size_t slash = self.rfind('/'); if (slash != std::string::npos) self = self.substr(slash + 1);
    std::string mode, name;
    {   // --batch FILE [--gpus N]   (and, for the ranks the command starts itself: --rank R --world N, results on descriptor 3)
        std::vector<std::string> a(argv + 1, argv + argc);
        std::string bmode = self.find("verifier") != std::string::npos ? "verifier" : "prover";
        if (!a.empty() && (a[0] == "prover" || a[0] == "verifier")) { bmode = a[0]; a.erase(a.begin()); }
        if (a.size() >= 2 && a[0] == "--batch") {
            try {
                uint32_t gpus = 1, rank = 0, world = 0, workers = 4;
                if (a.size() % 2) { std::fprintf(stderr, "option %s needs a value\n", a.back().c_str()); return 2; }
                for (size_t k = 2; k + 1 < a.size(); k += 2) {
                    if (a[k] == "--gpus") gpus = (uint32_t)std::stoul(a[k + 1]);
                    else if (a[k] == "--rank") rank = (uint32_t)std::stoul(a[k + 1]);
                    else if (a[k] == "--world") world = (uint32_t)std::stoul(a[k + 1]);
                    else if (a[k] == "--workers") workers = (uint32_t)std::stoul(a[k + 1]);
                    else { std::fprintf(stderr, "unknown option %s\n", a[k].c_str()); return 2; }
                }
                if (world) {
                    FILE *out = fdopen(3, "w");
                    if (!out) { std::fprintf(stderr, "rank %u: no result pipe\n", rank); return 101; }
                    const int rc = run_batch_rank(bmode, read_batch(a[1]), rank, world, workers, out);
                    std::fclose(out);
                    return rc;
                }
                if (gpus < 1 || gpus > 64 || workers < 1 || workers > 32) { std::fprintf(stderr, "--gpus 1..64, --workers 1..32\n"); return 2; }
                if (gpus > 1 && profiler_preloaded()) {
                    // --gpus N forks and re-executes this image per rank, which is safe only while this process has not touched the GPU; a profiler's
                    // preloaded library (rocprofv3) initialises the GPU before main(), and an exec from such a process takes the machine down
                    std::fprintf(stderr, "--gpus %u under a profiler: profile ONE rank instead (rocprofv3 ... -- bpg_prover --batch FILE, or --rank R --world N)\n", gpus);
                    return 2;
                }
                char exe[4096]; const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
                return run_batch(n > 0 ? std::string(exe, (size_t)n) : std::string(argv[0]), bmode, a[1], gpus, workers);
            } catch (const std::exception &e) {
                std::fprintf(stderr, "%s --batch: %s\n", bmode.c_str(), e.what());
                return 101;
            }
        }
    }
    if (argc == 3 && (std::string(argv[1]) == "prover" || std::string(argv[1]) == "verifier")) { mode = argv[1]; name = argv[2]; }
    else if (argc == 2 && self.find("verifier") != std::string::npos) { mode = "verifier"; name = argv[1]; }
    else if (argc == 2 && self.find("prover") != std::string::npos) { mode = "prover"; name = argv[1]; }
    else {
        std::fprintf(stderr, "usage: bpg_prover NAME | bpg_verifier NAME | %s prover|verifier NAME\n"
                             "       bpg_prover|bpg_verifier --batch FILE [--gpus N] [--workers W]     (one stem per line of FILE; a stem that fails is reported, the others are still proved;\n"
                             "                                                                          under a profiler use --gpus 1: the ranks of --gpus N are re-executions of this image)\n", self.c_str());
        return 2;
    }
    // a run makes ONE proof per stem: bpg_ctx_create's default, the one-shot profile (15 odd multiples, 7 ms to build at 2^20; no 8-bit tail tables)
    try {
        if (mode == "prover") { ProverRun r; r.name = name; return r.run(); }
        VerifierRun r; r.name = name; return r.run();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s: %s\n", mode.c_str(), e.what());
        return 101;                                                       // the reference unwrap()s: panic exit code
    }
}
#endif  // BPG_CLI_NO_MAIN
