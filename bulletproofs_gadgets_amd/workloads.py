"""Synthetic benchmark circuits of BASELINE.json's configs, assembled through the gadget surface exactly as the
reference's tests do (library path of SURVEY.md section 3.2).  Seeds: SHAKE256("bpg-synth" || cfg || index)."""
import hashlib

from . import (BoundsCheck, MerkleTree256, MimcHash256, Prover, Transcript, be_to_scalar, commit, commit_all_single,
               commit_single, hash_pattern, mimc_hash, scalar_to_be, vars_to_lc, L)


def synth(cfg: str, index: int, nbytes: int = 32) -> bytes:
    return hashlib.shake_256(b"bpg-synth" + cfg.encode() + index.to_bytes(4, "little")).digest(nbytes)


def blinding(cfg: str, index: int) -> bytes:
    """stands in for Scalar::random(&mut thread_rng()) (reference src/gadget.rs:31, src/commitments.rs:27,39)"""
    return (int.from_bytes(synth(cfg + "-blind", index, 64), "little") % L).to_bytes(32, "little")


def _setup(g, p, scalars, blindings):
    """Gadget::setup (reference src/gadget.rs:18-38); a prover that makes its commitments outside the library (the fixture generator of
    tests/golden, which has no device) brings its own."""
    return p.gadget_setup(g, scalars, blindings) if hasattr(p, "gadget_setup") else g.setup(p, scalars, blindings)


class Assembled:
    """A prover with every gadget assembled, plus what a verifier needs to rebuild the same statement."""
    def __init__(self, prover, transcript, commitments, gens_capacity, replay):
        self.prover, self.transcript, self.commitments, self.gens_capacity, self.replay = prover, transcript, commitments, gens_capacity, replay


def bounds_check_64(ctx, seed=0, label=b"BoundsCheck", prover_cls=Prover):
    """cfg 2: one BOUND W0 I0 I1 with 8-byte bounds (64-bit range proofs): n = N = 128, q = 259, m = 3."""
    cfg = "cfg2-%d" % seed
    lo, hi = bytes(8), b"\xff" * 8
    witness = synth(cfg, 0, 8)
    t = Transcript(label)
    p = prover_cls(ctx, t)
    g = BoundsCheck(lo, hi)
    scalars, wcoms, wvars = commit(p, witness, [blinding(cfg, 0)])
    dcoms, derived = _setup(g, p, scalars, [blinding(cfg, 1), blinding(cfg, 2)])
    g.prove(p, wvars, derived)

    def replay(v):
        from . import verifier_commit
        wv = verifier_commit(v, wcoms)
        dv = verifier_commit(v, dcoms)
        BoundsCheck(lo, hi).verify(v, wv, dv)
    return Assembled(p, t, wcoms + dcoms, 128, replay)


def mimc_preimage(ctx, nbytes=2130, seed=0, label=b"MiMCHash", prover_cls=Prover):
    """cfg 3: one HASH I0 W0 over a seeded preimage; 2130 bytes -> 67 absorbed blocks, n = 65,124, N = 2^16."""
    cfg = "cfg3-%d" % seed
    pre = synth(cfg, 0, nbytes)
    image = mimc_hash(pre)
    t = Transcript(label)
    p = prover_cls(ctx, t)
    g = MimcHash256(image)
    nblocks = (nbytes + 31) // 32
    scalars, wcoms, wvars = commit(p, pre, [blinding(cfg, i) for i in range(nblocks)])
    dcoms, derived = _setup(g, p, scalars, [blinding(cfg, 1000), blinding(cfg, 1001)])
    g.prove(p, wvars, derived)
    n = p.get_num_multiplications()
    cap = 1
    while cap < n:
        cap *= 2

    def replay(v):
        from . import verifier_commit
        wv = verifier_commit(v, wcoms)
        dv = verifier_commit(v, dcoms)
        MimcHash256(image).verify(v, wv, dv)
    return Assembled(p, t, wcoms + dcoms, cap, replay)


def full_tree_pattern(leaves: int) -> str:
    pat = "W"
    k = 1
    while k < leaves:
        pat = hash_pattern(pat, pat)
        k *= 2
    return pat


def merkle_full_tree(ctx, leaves=512, seed=None, label=b"MerkleTree", prover_cls=Prover):
    """cfg 4: MerkleTree256 over a full binary tree with every leaf a committed witness
    (reference src/merkle_tree/merkle_tree_gadget.rs:473-545: 512 leaves, n = 993,384, N = 2^20, m = 512).
    seed None reproduces the reference's instance (512 x leaf W1, root at :476); an integer derives distinct leaves."""
    cfg = "cfg4-%s" % seed
    if seed is None:
        leaf_be = [bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")] * leaves
    else:
        leaf_be = [b"\x07" + synth(cfg, i, 31) for i in range(leaves)]      # top byte small: canonical scalars
    t = Transcript(label)
    p = prover_cls(ctx, t)
    scalars, wcoms, wvars = commit_all_single(p, leaf_be, [blinding(cfg, i) for i in range(leaves)])
    pattern = full_tree_pattern(leaves)
    if seed is None and leaves == 512:
        root = be_to_scalar(bytes.fromhex("038c137beec8e2edfb5c48cbd063f04e569139d2221a4eb7befb85aa1bf8ba40"))   # merkle_tree_gadget.rs:476
    else:
        # the node sponge (mimc.rs:26-40) is private in the reference; let an assembly-only prover synthesise the tree
        # over instance leaves and read the root off the last multiplier output (p + k with k = 0)
        probe = Prover(None, Transcript(b"probe"))
        MerkleTree256(bytes(32), [be_to_scalar(b) for b in leaf_be], [], pattern.replace("W", "I")).prove(probe, [], [])
        root = probe.instance().aO[-32:]
    MerkleTree256(root, [], vars_to_lc(wvars), pattern).prove(p, [], [])
    n = p.get_num_multiplications()
    cap = 1
    while cap < n:
        cap *= 2

    def replay(v):
        from . import verifier_commit
        wv = verifier_commit(v, wcoms)
        MerkleTree256(root, [], vars_to_lc(wv), pattern).verify(v, [], [])
    a = Assembled(p, t, wcoms, cap, replay)
    a.root = root
    return a


def merkle_path_files(stem: str, depth=20, wbytes=16, seed=0):
    """cfg 4b: a depth-`depth` Merkle authentication path through the file driver,
    `MERKLE I0 ((..((W0 I1) I2)..) I{depth})` (reference src/bin/prover.rs:307-339). Writes stem.gadgets/.inst/.wtns.
    W0 (< 32 bytes: one absorbed block) is hashed by hash_witness, every sibling I_k by mimc_hash (prover.rs:160-200):
    n = depth*1944 + 972 multipliers."""
    from . import mimc_hash
    cfg = "cfg4b-%s" % seed
    w0 = synth(cfg, 0, wbytes)
    sib = [synth(cfg, k, 24) for k in range(1, depth + 1)]
    tree, pat = "W0", "I"
    for k in range(1, depth + 1):
        tree = "(%s I%d)" % (tree, k)
        pat = hash_pattern(pat, "I")
    probe = Prover(None, Transcript(b"probe"))
    MerkleTree256(bytes(32), [mimc_hash(w0)] + [mimc_hash(s) for s in sib], [], pat).prove(probe, [], [])
    root_le = probe.instance().aO[-32:]
    with open(stem + ".gadgets", "w") as f:
        f.write("MERKLE I0 %s\n" % tree)
    with open(stem + ".inst", "w") as f:
        f.write("I0 = 0x%s\n" % root_le[::-1].hex())
        for k, s in enumerate(sib, 1):
            f.write("I%d = 0x%s\n" % (k, s.hex()))
    with open(stem + ".wtns", "w") as f:
        f.write("W0 = 0x%s\n" % w0.hex())
    return depth * 1944 + 972


def merkle_tree_files(stem: str, leaves=256, wbytes=16, seed=0):
    """A full `leaves`-leaf MiMC Merkle tree through the FILE driver: `MERKLE I0 ((..(W0 W1)..) (..))` with every leaf a witness, each hashed
    by hash_witness (reference src/bin/prover.rs:160-190, 307-339): n = leaves * 972 + (leaves - 1) * 1944 multipliers (256 leaves: 744,552,
    padded to 2^20), 3 * leaves + leaves commitments.  Writes stem.gadgets/.inst/.wtns; returns n."""
    from . import mimc_hash
    cfg = "files-%s" % seed
    ws = [synth(cfg, k, wbytes) for k in range(leaves)]
    level = ["W%d" % k for k in range(leaves)]
    pat = ["I"] * leaves
    while len(level) > 1:
        level = ["(%s %s)" % (level[i], level[i + 1]) for i in range(0, len(level), 2)]
        pat = [hash_pattern(pat[i], pat[i + 1]) for i in range(0, len(pat), 2)]
    probe = Prover(None, Transcript(b"probe"))
    MerkleTree256(bytes(32), [mimc_hash(w) for w in ws], [], pat[0]).prove(probe, [], [])
    root_le = probe.instance().aO[-32:]
    with open(stem + ".gadgets", "w") as f:
        f.write("MERKLE I0 %s\n" % level[0])
    with open(stem + ".inst", "w") as f:
        f.write("I0 = 0x%s\n" % root_le[::-1].hex())
    with open(stem + ".wtns", "w") as f:
        for k, w in enumerate(ws):
            f.write("W%d = 0x%s\n" % (k, w.hex()))
    return leaves * 972 + (leaves - 1) * 1944
