// FUZZ TARGET (test infrastructure, not product): libFuzzer + ASan + UBSan over the native file drivers' readers - .gadgets / .inst / .wtns /
// .coms / .proof - and everything behind them that runs without a device: the line parser and tree grammar of csrc/cli_main.cpp, the C ABI
// (capi.hip), the constraint system and the seven gadgets (csrc/host/r1cs.hpp, gadgets.hpp), MiMC, Merlin.  The drivers run in assemble-only
// mode: prover on a device-less bpg_prover whose commitments are hash bytes (bpg_test_prover_stub_commitments), verifier up to (not including)
// verify().  An input is five sections separated by a line "====": gadgets, inst, wtns, coms, proof; make_corpus.py seeds the corpus with the
// reference's own test files (tests/golden/resources: byte-identical copies of /root/reference/tests/resources inputs) and the prover's outputs.
// A driver failure (exception -> "panic" exit code in the product) is an expected outcome; a sanitizer report or a crash is a finding.
#define BPG_CLI_NO_MAIN
#include "../../bulletproofs_gadgets_amd/csrc/cli_main.cpp"
#include <sys/stat.h>

namespace {
std::string g_dir;
void put(const std::string &path, const std::string &data) { std::ofstream f(path, std::ios::binary); f.write(data.data(), (std::streamsize)data.size()); }
}  // namespace

extern "C" int LLVMFuzzerInitialize(int *, char ***) {
    static std::string tmpl = std::string("/dev/shm/bpgfuzz") + "XXXXXX";
    const char *d = mkdtemp(&tmpl[0]);
    g_dir = d ? d : "/tmp";
    setenv("BPG_CLI_SEED", "fuzz", 1);
    setenv("BPG_CLI_RNG_SEED", "0000000000000000000000000000000000000000000000000000000000000000", 1);
    return 0;
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) {
    std::string all(reinterpret_cast<const char *>(data), size), part[5];
    size_t pos = 0;
    for (int k = 0; k < 5; k++) {
        const size_t sep = k < 4 ? all.find("\n====\n", pos) : std::string::npos;
        part[k] = all.substr(pos, sep == std::string::npos ? std::string::npos : sep - pos);
        if (sep == std::string::npos) break;
        pos = sep + 6;
    }
    const std::string stem = g_dir + "/f";
    put(stem + ".gadgets", part[0]); put(stem + ".inst", part[1]); put(stem + ".wtns", part[2]); put(stem + ".coms", part[3]); put(stem + ".proof", part[4]);
    for (int two_pass = 0; two_pass < 2; two_pass++) {
        setenv("BPG_CLI_TWO_PASS", two_pass ? "1" : "0", 1);
        try { ProverRun r; r.name = stem; r.assemble_only = true; r.quiet = true; (void)r.run(); } catch (const std::exception &) { }
    }
    try { VerifierRun v; v.name = stem; v.assemble_only = true; v.quiet = true; (void)v.run(); } catch (const std::exception &) { }
    return 0;
}
