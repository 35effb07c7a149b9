// host-only profile of the constraint assembly of the 512-leaf MiMC Merkle tree (diagnostic): g++ -O2 -pg, then gprof
#include <chrono>
#include <cstdio>
#include "../../bulletproofs_gadgets_amd/csrc/host/gadgets.hpp"
using namespace bpg;
namespace bpg {
class Engine {};                                                     // assembly needs no engine
std::vector<uint8_t> Prover::prove(uint64_t, const uint8_t *, uint32_t) { return {}; }
void Prover::start_blinding(const uint8_t *, uint64_t) {}
const std::vector<Scalar> &mimc_round_constants() {
    static const uint64_t RC[486][4] = {
#include "../../bulletproofs_gadgets_amd/csrc/host/mimc_rc769.inc"
    };
    static const std::vector<Scalar> v = [] { std::vector<Scalar> out(486); for (int i = 0; i < 486; i++) out[i] = Scalar::from_bits(reinterpret_cast<const uint8_t *>(RC[i])); return out; }();
    return v;
}
std::pair<std::vector<uint8_t>, Variable> Prover::commit(const Scalar &, const Scalar &) { return {{}, Variable::one()}; }
std::vector<Variable> Prover::commit_many(const std::vector<Scalar> &, const std::vector<Scalar> &, std::vector<uint8_t> &) { return {}; }
}
int main() {
    std::string pat = "I";
    for (int k = 1; k < 512; k *= 2) pat = "(" + pat + " " + pat + ")";
    uint8_t leaf_be[32]; const char *hex = "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749";
    for (int i = 0; i < 32; i++) { unsigned v; sscanf(hex + 2 * i, "%2x", &v); leaf_be[31 - i] = (uint8_t)v; }
    Scalar leaf = Scalar::from_bits(leaf_be);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        Transcript t("probe"); Prover p(nullptr, &t);
        std::vector<LinearCombination> inst(512, LinearCombination(leaf));
        MerkleTree256 g(LinearCombination(Scalar::zero()), inst, {}, Pattern::parse(pat));
        g.assemble(p, {}, {});
        auto t1 = std::chrono::steady_clock::now();
        FlatCircuit f = p.flatten();
        auto t2 = std::chrono::steady_clock::now();
        printf("assembly %.3f s  flatten %.3f s  n=%zu q=%zu\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), (size_t)f.n, f.row_ptr.size() - 1);
    }
    return 0;
}
