"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libbpg_hip.so and is
compared byte for byte with the CPU oracle on the same seeded inputs, then checked by the oracle's verifier."""
import hashlib
import pytest
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
H = bytes.fromhex
sc = lambda x: (x % R.L).to_bytes(32, "little")
rs = lambda tag, i: sc(int.from_bytes(hashlib.sha512(b"%s%d" % (tag, i)).digest(), "little"))


@pytest.fixture(scope="module")
def ctx():
    return bpg.Context(0)


def to_oracle(inst):
    return O.FlatCircuit(inst.n, inst.m, inst.aL or None, inst.aR or None, inst.aO or None, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)


def check_against_oracle(ctx, prover, transcript, commitments, capacity, replay=None, flags_list=(0,), seed=bytes(range(32)), ogens=None):
    """prove on the GPU for each dialect; compare with the oracle; verify with the oracle on the VERIFIER-side circuit."""
    inst = prover.instance()
    state = transcript.state
    oc = to_oracle(inst)
    assert O.satisfied(oc, inst.v)
    ogens = ogens or O.Gens(capacity)
    ctx.gens_ensure(capacity)
    res = ctx.upload(inst)
    vstate, vcirc = None, None
    if replay is not None:
        tv = bpg.Transcript(transcript._label)
        v = bpg.Verifier(tv)
        replay(v)
        vi = v.instance()
        assert vi.commitments == b"".join(commitments)
        vstate, vcirc = tv.state, to_oracle(vi)
        assert vstate == state
    proofs = []
    for flags in flags_list:
        proof, st_after = res.prove(state, inst.v_blinding, seed, flags)
        rc, want, st_want = O.prove(ogens, state, oc, inst.v_blinding, seed, flags | O.FLAG_FAST_MSM)
        assert rc == 0
        assert proof == want, "flags=%d: GPU proof bytes differ from the oracle" % flags
        assert st_after == st_want
        coms = b"".join(commitments)
        assert O.verify(ogens, vstate or state, vcirc or oc, coms, proof, flags=flags) == 0
        # the GPU verifier (bpg_r1cs_verify) must take the same decisions as the oracle verifier
        vinst = vi if replay is not None else inst
        assert ctx.verify_flat(vinst, vstate or state, coms, proof, flags=flags) == 0
        assert res.verify(vstate or state, coms, proof, flags=flags) == 0           # same decision on the resident (prover-side) upload
        assert res.verify(vstate or state, coms, proof[:40] + bytes([proof[40] ^ 1]) + proof[41:], flags=flags) in (2, 3)
        for pos in (5, 100, len(proof) - 40, len(proof) - 1):
            bad = bytearray(proof); bad[pos] ^= 0x01
            want = O.verify(ogens, vstate or state, vcirc or oc, coms, bytes(bad), flags=flags)
            got = ctx.verify_flat(vinst, vstate or state, coms, bytes(bad), flags=flags)
            assert want != 0 and got == want, (pos, want, got)
        assert ctx.verify_flat(vinst, vstate or state, coms, proof[:-1], flags=flags) == 2
        if coms:
            badc = bytearray(coms); badc[0] ^= 2
            # rejected unless the circuit never uses that commitment (random circuits): the decision must be the oracle's
            want = O.verify(ogens, vstate or state, vcirc or oc, bytes(badc), proof, flags=flags)
            got = ctx.verify_flat(vinst, vstate or state, bytes(badc), proof, flags=flags)
            assert (got == 0) == (want == 0) and got in (0, 2, 3)
        proofs.append(proof)
    res.free()
    return proofs


class LabeledTranscript(bpg.Transcript):
    def __init__(self, label):
        super().__init__(label)
        self._label = label


def test_device_field_ops(ctx):
    """the gfx950 inline-asm field arithmetic (fe.cuh device paths) against Python big integers, incl. weakly reduced edge values"""
    P = R.P
    edge = [0, 1, 2, 19, 37, 38, 39, P - 1, P, P + 1, 2 * P - 1, 2 * P, 2 * P + 1, 2**255 - 1, 2**255, 2**256 - 39, 2**256 - 38, 2**256 - 1,
            2**256 - 2**32, 2**224, (2**256 - 1) ^ (2**128 - 1), 2**32 - 1, 2**64 - 1, (2**256 - 1) // 3]
    vals = edge + [int.from_bytes(hashlib.sha256(b"fe%d" % i).digest(), "little") for i in range(400)]
    A = [v.to_bytes(32, "little") for v in vals for _ in range(3)]
    Bv = []
    for i, v in enumerate(vals):
        Bv += [vals[(7 * i + 3) % len(vals)], vals[(13 * i + 5) % len(vals)], edge[i % len(edge)]]
    B = [v.to_bytes(32, "little") for v in Bv]
    av = [int.from_bytes(x, "little") for x in A]
    ops = {0: lambda a, b: a * b, 1: lambda a, b: a * a, 2: lambda a, b: a + b, 3: lambda a, b: a - b}
    for op, f in ops.items():
        got = ctx.test_fe_ops(op, A, B)
        for a, b, g in zip(av, Bv, got):
            assert int.from_bytes(g, "little") == f(a, b) % P, (op, hex(a), hex(b))
    got = ctx.test_fe_ops(4, A[:120], B[:120])
    for a, g in zip(av[:120], got):
        assert int.from_bytes(g, "little") == pow(a % P, P - 2, P)
    got = ctx.test_fe_ops(5, A[:200], B[:200])
    for a, b, g in zip(av[:200], Bv[:200], got):
        x, y = a % P, b % P
        for _ in range(25):
            t = (x * y - (x + y)) % P
            x = (y - t) ** 2 % P
            y = (t - x) % P
        assert int.from_bytes(g, "little") == (x + y) % P
    # op 6: the scalar field's Montgomery product (sc.cuh device path: product scanning + Montgomery reduction on l's five non-zero limbs);
    # contract: one operand below l, the other any 256-bit value
    Lq = R.L
    Rinv = pow(2**256, -1, Lq)
    sedge = [0, 1, 2, Lq - 1, Lq - 2, 2**252, 2**252 - 1, 2**128, (Lq - 1) // 2, 2**32 - 1, 2**64, 0x0fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff]
    sa = sedge + [int.from_bytes(hashlib.sha256(b"sa%d" % i).digest(), "little") % Lq for i in range(300)]
    sb = [2**256 - 1, 2**256 - 2**32, Lq, Lq + 1, 2 * Lq, 0, 1, Lq - 1, 2**255, 2**253, 3, 2**256 - Lq] + [int.from_bytes(hashlib.sha256(b"sb%d" % i).digest(), "little") for i in range(300)]
    got = ctx.test_fe_ops(6, [v.to_bytes(32, "little") for v in sa], [v.to_bytes(32, "little") for v in sb])
    for x, y, g in zip(sa, sb, got):
        assert int.from_bytes(g, "little") == x * y * Rinv % Lq, (hex(x), hex(y))


def test_pedersen_bases_and_commitments(ctx, golden):
    B, Bb = ctx.pedersen_bases()
    assert B.hex() == golden["pedersen"]["B"] and Bb.hex() == golden["pedersen"]["B_blinding"]
    vs = [H(v["v"]) for v in golden["pedersen_commit"]]
    rr = [H(v["r"]) for v in golden["pedersen_commit"]]
    assert [c.hex() for c in ctx.pedersen_commit(vs, rr)] == [v["commit"] for v in golden["pedersen_commit"]]
    vs = [rs(b"pv", i) for i in range(130)] + [bytes(32), b"\xff" * 31 + b"\x7f"]
    rr = [rs(b"pr", i) for i in range(130)] + [bytes(32), sc(R.L - 1)]
    got = ctx.pedersen_commit(vs, rr)
    for v, r, g in zip(vs, rr, got):
        assert g == O.pedersen_commit(v, r)


def test_generators_match_golden_and_oracle(ctx, golden):
    ctx.gens_ensure(2048)
    G, Hh = ctx.gens_export(0, 8)
    assert [G[32 * i:32 * i + 32].hex() for i in range(8)] == golden["bp_gens"]["G"]
    assert [Hh[32 * i:32 * i + 32].hex() for i in range(8)] == golden["bp_gens"]["H"]
    og = O.Gens(2048)
    assert ctx.gens_export(0, 2048) == og.export(0, 2048)
    ctx.gens_ensure(4096)                      # growing keeps the prefix (GeneratorsChain is a stream)
    assert ctx.gens_export(1000, 1048) == og.export(1000, 1048)
    with pytest.raises(bpg.BpgError):
        ctx.gens_ensure(1000)                  # not a power of two


@pytest.mark.parametrize("count", [1, 2, 3, 33, 190, 700, 5000])
def test_msm_matches_oracle(ctx, count):
    ctx.gens_ensure(8192)
    G, Hh = ctx.gens_export(7, count)
    s = [rs(b"ms", i) for i in range(count)]
    t = [rs(b"mt", i) for i in range(count)]
    # edge scalars: zeros, ones (everything lands in one bucket), l-1
    for i in range(0, count, 5):
        s[i] = bytes(32)
    for i in range(1, count, 7):
        t[i] = sc(1)
    if count > 2:
        s[2] = sc(R.L - 1)
    want = O.msm(b"".join(s + t), G + Hh, 1)
    assert ctx.msm_gens(7, s, t) == want


@pytest.mark.parametrize("cmin", [16, 15, 9])
def test_msm_with_wide_windows_on_small_sums(cmin, monkeypatch):
    """Windows of up to 16 bits (what a 2^20-term sum takes while other proofs share the device: 16 windows instead of 17): the stored digit is
    digit + 2^15 in 16 bits, -2^15 and the empty digit included.  Forced here on sums small enough for the oracle (BPG_MSM_CMIN)."""
    monkeypatch.setenv("BPG_MSM_CMIN", str(cmin))
    c = bpg.Context(0)
    try:
        c.gens_ensure(8192)
        for count in (3, 700, 5000):
            G, Hh = c.gens_export(5, count)
            s = [rs(b"ws", i) for i in range(count)]
            t = [rs(b"wt", i) for i in range(count)]
            for i in range(0, count, 5):
                s[i] = bytes(32)
            for i in range(1, count, 7):
                t[i] = sc(1)
            # digits at both ends of a 16-bit window: ...8000 (-2^15 after the recoding), ...7fff, ...ffff, and the same one window up
            edge = [0x8000, 0x7fff, 0xffff, 0x8000 << 16, 0x7fff << 16, (1 << 253) - 1, R.L - 1, 0x80008000800080008000]
            for k, v in enumerate(edge):
                if 2 + k < count:
                    s[2 + k] = sc(v % R.L)
            assert c.msm_gens(5, s, t) == O.msm(b"".join(s + t), G + Hh, 1), (cmin, count)
    finally:
        c.close()


@pytest.mark.parametrize("env", [{"BPG_RSEG": "1"}, {"BPG_RSEG": "2"}, {"BPG_RSEG": "64"}, {"BPG_RSEG": "1024"}, {"BPG_LGCH": "2"}, {"BPG_LGCH": "7"},
                                 {"BPG_SWEEP_RESIDENT": "64"}, {"BPG_SWEEP_RESIDENT": "4096", "BPG_FOLD_ADAPT": "2"}, {"BPG_FOLD_ADAPT": "2", "BPG_RSEG": "16"},
                                 {"BPG_FOLD_ADAPT": "0", "BPG_MSM_CMAX": "12"},
                                 {"BPG_WINDOW_QUAD": "0"}, {"BPG_WINDOW_QUAD_BLOCKS": "1"}, {"BPG_WINDOW_QUAD_BLOCKS": "65536"},
                                 {"BPG_WINDOW_QUAD_BLOCKS": "65536", "BPG_RSEG": "1"}, {"BPG_WINDOW_QUAD_BLOCKS": "40", "BPG_RSEG": "2"}])
def test_msm_epilogue_and_sweep_knobs_give_the_oracle_sum(env, monkeypatch):
    """The diagnostic knobs of the bucket-method MSM - BPG_RSEG (buckets per thread of the first level of the window epilogue), BPG_LGCH (chunk
    length of the sweep), BPG_SWEEP_RESIDENT (blocks the device is taken to hold), BPG_FOLD_ADAPT = 2 (the shared-device variants: 16-bit windows,
    64-entry chunks, 256-thread window blocks), BPG_WINDOW_QUAD / BPG_WINDOW_QUAD_BLOCKS (the window sums of a proof alone with four lanes per point:
    off; one block per window; up to 64 blocks per window, whose last one runs the second stage on one or on four waves) - change how the sum is
    scheduled, never its value: windows from 2 to 16 bits wide, with and without whole empty segments, identical scalars in one bucket, against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for cmin in ("2", "11", "16"):
        monkeypatch.setenv("BPG_MSM_CMIN", cmin)
        c = bpg.Context(0)
        try:
            c.gens_ensure(8192)
            for count in (1, 70, 3000):
                G, Hh = c.gens_export(3, count)
                s = [rs(b"ks", i) for i in range(count)]
                t = [rs(b"kt", i) for i in range(count)]
                for i in range(0, count, 3):
                    t[i] = sc(R.L - 2)                          # a third of the terms in ONE bucket of every window (the heavy list)
                for i in range(1, count, 11):
                    s[i] = bytes(32)
                for i in range(2, count, 75):
                    s[i] = sc(0x123456789abcdef0fedcba9876543211)   # up to 40 equal scalars: a bucket over a handful of chunks (the medium list of k_bucket_combine)
                assert c.msm_gens(3, s, t) == O.msm(b"".join(s + t), G + Hh, 1), (env, cmin, count)
        finally:
            c.close()


@pytest.mark.parametrize("merge", ["1", "0", "2"])
def test_equal_scalar_merging_changes_no_byte(merge, monkeypatch):
    """Terms of A_I that carry the same scalar share one bucket entry per window on the sum of their generators (hip/k_merge.cuh; a MiMC round wires
    a_L[2i] = a_R[2i] = a_R[2i+1], reference src/mimc_hash/mimc_hash_gadget.rs:133-144): grouped by value once per uploaded witness.  With the merging
    on and off (BPG_MERGE), on circuits with groups of two, three, eight, nine, forty and hundreds of members (range-proof bits; a value class beyond 32
    members becomes several groups) and with zero scalars (never grouped), the proof is the oracle's, byte for byte.  BPG_TT_LG=0 sends these small circuits down the
    bucket-method path that the full-size ones take."""
    monkeypatch.setenv("BPG_MERGE", merge)
    monkeypatch.setenv("BPG_TT_LG", "0")
    c = bpg.Context(0)
    try:
        a = workloads.mimc_preimage(c, nbytes=100, seed=21, label=b"MiMCHash")
        check_against_oracle(c, a.prover, a.transcript, a.commitments, a.gens_capacity, flags_list=(0, 1))
        sch = c.schedule()
        assert sch["merge_equal"] == int(merge)                                   # (2: grouped afresh in every proof)
        n = a.prover.get_num_multiplications()
        assert sch["merged_last"] == (0 if merge == "0" else n // 2), sch        # one group {a_L[2i], a_R[2i], a_R[2i+1]} per MiMC round of A_I; a_O = t^2, t^3: none
        assert sch["merged_skipped_last"] == (0 if merge == "0" else 3 * (n // 2))
        a = workloads.bounds_check_64(c, seed=22)                                # hundreds of equal bits: a few large groups
        check_against_oracle(c, a.prover, a.transcript, a.commitments, a.gens_capacity)
        # a hand-made witness: multipliers (x, y) with x drawn from a few values so that the value classes have 2, 3, 8, 9 and 40 members, some zero
        t = bpg.Transcript(b"merge")
        p = bpg.Prover(c, t)
        vals = [rs(b"mv", k) for k in range(6)]
        sizes = {0: 2, 1: 3, 2: 8, 3: 9, 4: 40}
        for k, cnt in sizes.items():
            for j in range(cnt):
                p.allocate_multiplier((vals[k], rs(b"my", 100 * k + j)))           # a_L = the class value, a_R distinct
        for j in range(5):
            p.allocate_multiplier((bytes(32), rs(b"mz", j)))                       # zero scalars: no entries, never a group
        l, r, o = p.allocate_multiplier((vals[5], vals[5]))                        # a_L[i] == a_R[i]: a group across the two halves
        p.constrain(bpg.LinearCombination.of(l) - r)                               # (one constraint, satisfied: left = right)
        check_against_oracle(c, p, t, [], 128)
        if merge != "0":
            assert c.schedule()["merged_last"] == 7                                # the five value classes of a_L (the one of 40 members as two groups) and the (x, x) pair; a_O = x y: all different
    finally:
        c.close()


def test_one_context_interleaves_sizes_and_entry_points():
    """The large buffers of a context share ONE arena (engine.hip: an MSM's digits and entry lists, the sweep's partial sums, the polynomial phase's weights and
    powers, the tail's window tables take it in turn) and the window tables of the ORIGINAL generators of a small circuit live beside it and outlive the proof.
    On one context: a 2^16 proof (bucket-method sums, tail tables in the arena), its verification (one MSM over the arena), a 2^12 proof (table-driven from round
    0: tables of its own), a raw MSM, the 2^16 proof again, the small one again, an upload in between - every result equals the oracle's or the first run's."""
    c = bpg.Context(0)
    try:
        big = workloads.mimc_preimage(c, nbytes=2130, seed=31, label=b"MiMCHash")            # n = 65,124, N = 2^16
        small = workloads.mimc_preimage(c, nbytes=100, seed=32, label=b"MiMCHash")           # n = 3,888, N = 2^12
        bi, bs = big.prover.instance(), big.transcript.state
        si, ss = small.prover.instance(), small.transcript.state
        c.gens_ensure(big.gens_capacity)
        rb, rs_ = c.upload(bi), c.upload(si)
        seed = bytes([7]) * 32
        rc, want_small, _ = O.prove(O.Gens(small.gens_capacity), ss, to_oracle(si), si.v_blinding, seed, O.FLAG_FAST_MSM)
        assert rc == 0
        first_big = rb.prove(bs, bi.v_blinding, seed, 0)[0]
        assert O.verify(O.Gens(big.gens_capacity), bs, to_oracle(bi), b"".join(big.commitments), first_big) == 0
        G, Hh = c.gens_export(5, 300)
        s = [rs(b"is", i) for i in range(300)]; t = [rs(b"it", i) for i in range(300)]
        want_msm = O.msm(b"".join(s + t), G + Hh, 1)
        for rep in range(2):
            assert rb.verify(bs, b"".join(big.commitments), first_big) == 0
            assert rs_.prove(ss, si.v_blinding, seed, 0)[0] == want_small
            assert c.msm_gens(5, s, t) == want_msm
            assert rb.prove(bs, bi.v_blinding, seed, 0)[0] == first_big
            assert rs_.verify(ss, b"".join(small.commitments), want_small) == 0
            r2 = c.upload(bi)                                                               # an upload's workspace is the arena too
            assert r2.prove(bs, bi.v_blinding, seed, 0)[0] == first_big
            r2.free()
            assert rs_.prove(ss, si.v_blinding, seed, 0)[0] == want_small
        rb.free(); rs_.free()
    finally:
        c.close()


def test_deferred_commitments_equal_one_launch_per_commitment(ctx):
    """bpg_prover_defer_commitments (batched witness commitment): the variables are registered at once, the commitments come
    from ONE k_pedersen launch at the flush and enter the transcript in commit order - same commitments, same transcript, same proof as the
    reference's one Prover::commit at a time (src/gadget.rs:27-35, src/lalrpop/assignment_parser.rs:152-169)."""
    class Deferring(bpg.Prover):
        def __init__(self, c, t):
            super().__init__(c, t)
            self.defer_commitments(True)

    for build in (lambda cls: workloads.mimc_preimage(ctx, nbytes=200, seed=11, label=b"MiMCHash", prover_cls=cls),
                  lambda cls: workloads.bounds_check_64(ctx, seed=12, prover_cls=cls),
                  lambda cls: workloads.merkle_full_tree(ctx, leaves=4, seed=13, prover_cls=cls)):
        plain, late = build(bpg.Prover), build(Deferring)
        m = len(plain.commitments)
        if m:
            assert late.commitments == [bytes(32)] * m                      # not known yet
            with pytest.raises(bpg.BpgError):
                late.prover.commitment(0)
            assert late.transcript.state != plain.transcript.state
        late.prover.flush_commitments()
        assert [late.prover.commitment(i) for i in range(m)] == plain.commitments
        assert late.transcript.state == plain.transcript.state
        with pytest.raises(bpg.BpgError):
            late.prover.commitment(m)
        # a commitment made after the flush goes into the next batch; turning deferral off flushes it
        extra = late.prover.commit(rs(b"dv", 1), rs(b"db", 1))
        assert extra[0] == bytes(32)
        late.prover.defer_commitments(False)
        want = plain.prover.commit(rs(b"dv", 1), rs(b"db", 1))
        assert late.prover.commitment(m) == want[0] and late.transcript.state == plain.transcript.state
        seed = bytes(range(32))
        gens = bpg.BulletproofGens(ctx, max(plain.gens_capacity, 2))
        assert late.prover.prove(gens, seed) == plain.prover.prove(gens, seed)
    # prove() flushes what is pending
    t1, t2 = bpg.Transcript(b"pend"), bpg.Transcript(b"pend")
    p1, p2 = bpg.Prover(ctx, t1), bpg.Prover(ctx, t2)
    p2.defer_commitments(True)
    for p in (p1, p2):
        _, (a, b, c) = p.commit_many([rs(b"pv", i) for i in range(3)], [rs(b"pb", i) for i in range(3)])
        l, r, o = p.multiply(a, b)
        p.constrain(bpg.LinearCombination.of(o) - o)
    assert p1.prove(bpg.BulletproofGens(ctx, 2), bytes(32)) == p2.prove(bpg.BulletproofGens(ctx, 2), bytes(32))
    assert [p2.commitment(i) for i in range(3)] == [p1.commitment(i) for i in range(3)]


def test_msm_all_ones_and_all_zero(ctx):
    ctx.gens_ensure(8192)
    n = 3000
    G, Hh = ctx.gens_export(0, n)
    ones, zeros = [sc(1)] * n, [bytes(32)] * n
    assert ctx.msm_gens(0, ones, zeros) == O.msm(b"".join(ones), G, 1)          # a single heavy bucket
    assert ctx.msm_gens(0, zeros, zeros) == bytes(32)                            # identity encodes as zeros


def test_tiny_circuits_all_dialects(ctx):
    # n = 1, 2, 3, 5: padded sizes 1, 2, 4, 8; committed x, y, z with x*y = z
    for nm in (1, 2, 3, 5):
        t = LabeledTranscript(b"unit")
        p = bpg.Prover(ctx, t)
        coms = []
        for i in range(nm):
            x, y = rs(b"x", i), rs(b"y", i)
            z = bpg.scalar_op("mul", x, y)
            cs, vs = p.commit_many([x, y, z], [rs(b"bl", 3 * i + k) for k in range(3)])
            coms += cs
            l, r, o = p.multiply(vs[0], vs[1])
            p.constrain(bpg.LinearCombination.of(o) - vs[2])
        assert p.get_num_multiplications() == nm and p.num_constraints() == 3 * nm
        check_against_oracle(ctx, p, t, coms, 16, flags_list=(0, 1, 2, 3))


def test_constraint_only_circuit_n0(ctx):
    # no multipliers at all (e.g. the reference's EQUALS gadget): n = 0 pads to N = 1
    t = LabeledTranscript(b"eq")
    p = bpg.Prover(ctx, t)
    x = rs(b"eqv", 0)
    cs, vs = p.commit_many([x, x], [rs(b"eqb", 0), rs(b"eqb", 1)])
    p.constrain(bpg.LinearCombination.of(vs[0]) - vs[1])
    check_against_oracle(ctx, p, t, cs, 1)


def test_range_proof_reference_cases(ctx):
    # reference src/utils.rs:46-90: constant LC, no commitments (m = 0); n = 56 verifies, n = 48 must not
    x = bpg.be_to_scalar(H("0522a64d7b931e"))
    for nbits, ok in ((56, True), (48, False)):
        t = bpg.Transcript(b"RangeProof")
        p = bpg.Prover(ctx, t)
        bpg.range_proof(p, x, nbits, x)
        inst = p.instance()
        state = t.state
        proof = p.prove(bpg.BulletproofGens(ctx, 256), bytes(32))
        og = O.Gens(256)
        rc, want, _ = O.prove(og, state, to_oracle(inst), b"", bytes(32), O.FLAG_FAST_MSM)
        assert proof == want
        tv = bpg.Transcript(b"RangeProof")
        v = bpg.Verifier(tv)
        bpg.range_proof(v, x, nbits)
        assert (O.verify(og, tv.state, to_oracle(v.instance()), b"", proof) == 0) == ok


def test_bounds_check_gadget_reference_case(ctx):
    # reference src/bounds_check/bounds_check_gadget.rs:75-99
    lo, hi, w = bytes([10]), bytes([100]), bytes([67])
    t = LabeledTranscript(b"BoundsCheck")
    p = bpg.Prover(ctx, t)
    g = bpg.BoundsCheck(lo, hi)
    scalars, wc, wv = bpg.commit(p, w, [rs(b"bc", 0)])
    dc, dw = g.setup(p, scalars, [rs(b"bc", 1), rs(b"bc", 2)])
    g.prove(p, wv, dw)

    def replay(v):
        g2 = bpg.BoundsCheck(lo, hi)
        g2.verify(v, bpg.verifier_commit(v, wc), bpg.verifier_commit(v, dc))
    check_against_oracle(ctx, p, t, wc + dc, 16, replay, flags_list=(0, 1))
    # out-of-range witness: proof is produced but must not verify
    t = LabeledTranscript(b"BoundsCheck")
    p = bpg.Prover(ctx, t)
    scalars, wc, wv = bpg.commit(p, bytes([101]), [rs(b"bc", 0)])
    dc, dw = g.setup(p, scalars, [rs(b"bc", 1), rs(b"bc", 2)])
    g.prove(p, wv, dw)
    inst, state = p.instance(), t.state
    proof = p.prove(16, bytes(32))
    assert O.verify(O.Gens(16), state, to_oracle(inst), b"".join(wc + dc), proof) != 0
    with pytest.raises(bpg.BpgError) as e:      # capacity below the padded size -> InvalidGeneratorsLength
        t2 = LabeledTranscript(b"BoundsCheck"); p2 = bpg.Prover(ctx, t2)
        s2, _, v2 = bpg.commit(p2, w, [rs(b"bc", 0)]); _, d2 = g.setup(p2, s2, [rs(b"bc", 1), rs(b"bc", 2)]); g.prove(p2, v2, d2)
        p2.prove(8, bytes(32))
    assert e.value.status == 1


def test_cfg2_bounds_check_64(ctx):
    a = workloads.bounds_check_64(ctx, seed=3, label=b"BoundsCheck")
    a.transcript._label = b"BoundsCheck"
    inst = a.prover.instance()
    assert (inst.n, inst.q, inst.m) == (128, 259, 3)
    check_against_oracle(ctx, a.prover, a.transcript, a.commitments, 128, a.replay, flags_list=(0, 1, 2, 3))


@pytest.mark.parametrize("pre,label", [(H("38535450433043546f313877615a6a423663"), b"MiMCHash"),
                                        (b"The quick brown fox jumps over t", b"MiMCHash"),
                                        (H("546865207175694a76077d4a40bd91551b3a03b1ad8adb2b666f78206a756d70666f78206a756d7073206f7665"), b"MiMCHash")])
def test_mimc_hash_gadget_reference_cases(ctx, pre, label):
    # reference src/mimc_hash/mimc_hash_gadget.rs:163-272 (test_mimc_hash_gadget_1..3: padding case, extra-block edge case, 45-byte preimage)
    image = bpg.mimc_hash(pre)
    t = LabeledTranscript(label)
    p = bpg.Prover(ctx, t)
    g = bpg.MimcHash256(image)
    scalars, wc, wv = bpg.commit(p, pre, [rs(b"mh", i) for i in range(4)])
    dc, dw = g.setup(p, scalars, [rs(b"mh", 10), rs(b"mh", 11)])
    g.prove(p, wv, dw)

    def replay(v):
        bpg.MimcHash256(image).verify(v, bpg.verifier_commit(v, wc), bpg.verifier_commit(v, dc))
    check_against_oracle(ctx, p, t, wc + dc, 2048, replay)
    # wrong image must not verify
    t = LabeledTranscript(label)
    p = bpg.Prover(ctx, t)
    bad = bpg.MimcHash256(bpg.mimc_hash(pre + b"x"))
    scalars, wc, wv = bpg.commit(p, pre, [rs(b"mh", i) for i in range(4)])
    dc, dw = bad.setup(p, scalars, [rs(b"mh", 10), rs(b"mh", 11)])
    bad.prove(p, wv, dw)
    inst, state = p.instance(), t.state
    proof = p.prove(2048, bytes(32))
    assert O.verify(O.Gens(2048), state, to_oracle(inst), b"".join(wc + dc), proof) != 0


W = {k: H(v) for k, v in {
    1: "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749", 2: "07faf8aaa21077200a11576b1cdb402f52a47f192b36998b4da25807a9be52f5",
    3: "09243333e374e76e4975ab48ae38241ba67805cd60f1523e9b79a48daac9a84d", 4: "0258647e47e8005748d4e7d0d76b230cc20f2a0f8745eee2bccced0c2add59d5",
    5: "011c6fc7f15087f4d3e97e672813af066f74f60446bc75aa85eb2d6db8ae791b", 6: "0f8653b7e734422fc75bdb4eb1bc774cd34f9ab3a89545e021016a4d9171a902",
    7: "0bd752eb80bfa5189bade1cc8f49cf5fe1843e1ff736367afc52670e429d1c36", 8: "181c63cfc823a477b0825004475222e1c7d060179b6b247ffa5adc58e307de0d",
    9: "2ad84a04eb9394e0cc4b4b478f211a815f2707597c6032a98a573fbdee4a3109", 10: "c45a435f3c401eeb6d3a08b2f93669ee33e4ad2640e4e9a9a34937006ae8b308",
    11: "acb33246c69545225a61fb60b44868e8bc8d25533c663aacabe449686bbed40c", 12: "7f7eba68d7be6b7076c17b6dc473a6d1770bcf1cb4266e7fb1e4642658050609",
    13: "a84d1ceceb0ebc710ba2bc5ae60bb6c38abad15f650bf7e87cb901533125110d", 14: "157cdbdece96312986c9f44e03c232d4ca9aad55e4e259828f1ac451a93dd40a",
    15: "a32f318c922b6404d6dd8eb2f65a73b05a49f14cb0b13f4828a840079e60460d"}.items()}


@pytest.mark.parametrize("pattern,wit,inst", [
    ("(((W W) (W W)) ((W W) (W W)))", [8, 9, 10, 11, 12, 13, 14, 15], []),          # merkle_tree_gadget.rs:218-258
    ("(((W W) (I W)) ((I W) (W I)))", [8, 9, 11, 13, 14], [10, 12, 15]),            # :260-304
    ("(((W W) (W W)) (W W))", [8, 9, 10, 11, 6, 7], []),                            # :306-345
    ("(((W W) (W W)) W)", [8, 9, 10, 11, 3], []),                                   # :347-385
    ("((W W) ((W W) (W W)))", [4, 5, 12, 13, 14, 15], []),                          # :387-427
    ("(W ((W W) (W W)))", [2, 12, 13, 14, 15], []),                                 # :429-468
])
def test_merkle_tree_gadget_reference_cases(ctx, pattern, wit, inst):
    root = bpg.be_to_scalar(W[1])
    t = LabeledTranscript(b"MerkleTree")
    p = bpg.Prover(ctx, t)
    ivals = [bpg.be_to_scalar(W[i]) for i in inst]
    _, wc, wv = bpg.commit_all_single(p, [W[i] for i in wit], [rs(b"mk", i) for i in range(len(wit))])
    bpg.MerkleTree256(root, ivals, bpg.vars_to_lc(wv), pattern).prove(p, [], [])

    def replay(v):
        bpg.MerkleTree256(root, ivals, bpg.vars_to_lc(bpg.verifier_commit(v, wc)), pattern).verify(v, [], [])
    check_against_oracle(ctx, p, t, wc, 16384, replay)


def test_combine_gadgets(ctx):
    # reference tests/combine_gadgets.rs:22-107: bounds + hash + merkle in one proof, 8192 generators
    t = LabeledTranscript(b"CombinedGadgets")
    p = bpg.Prover(ctx, t)
    w1_scalar, w1_com, w1_var = bpg.commit(p, bytes([67]), [rs(b"cg", 0)])
    image = H("0cfb0c17618211c607febf703ac3f3078f7d96798fae9d4a1682bc592f7cb126")
    _, w2_com, w2_var = bpg.commit_single(p, image, rs(b"cg", 1))
    lo, hi = bytes([17]), bytes([100])
    bounds = bpg.BoundsCheck(lo, hi)
    bdc, bdw = bounds.setup(p, w1_scalar, [rs(b"cg", 2), rs(b"cg", 3)])
    bounds.prove(p, w1_var, bdw)
    hsh = bpg.MimcHash256(w2_var)
    hdc, hdw = hsh.setup(p, w1_scalar, [rs(b"cg", 4), rs(b"cg", 5)])
    hsh.prove(p, w1_var, hdw)
    root = bpg.be_to_scalar(H("0c8c87b648e8fa0d9726ee8225be0628794f2e1d1ab932421d45851a35d81ac1"))
    leaf = bpg.be_to_scalar(W[3])
    bpg.MerkleTree256(root, [leaf], [w2_var], "(W I)").prove(p, [], [])

    def replay(v):
        wv = bpg.verifier_commit(v, w1_com + [w2_com])
        bpg.BoundsCheck(lo, hi).verify(v, [wv[0]], bpg.verifier_commit(v, bdc))
        bpg.MimcHash256(wv[1]).verify(v, [wv[0]], bpg.verifier_commit(v, hdc))
        bpg.MerkleTree256(root, [leaf], [wv[1]], "(W I)").verify(v, [], [])
    check_against_oracle(ctx, p, t, w1_com + [w2_com] + bdc + hdc, 8192, replay)


def test_cfg3_mimc_preimage_2_16(ctx):
    """BASELINE.json config 3: one HASH over a 2,130-byte preimage -> 67 absorbed blocks, n = 65,124, N = 2^16, q = 130,250."""
    a = workloads.mimc_preimage(ctx, nbytes=2130, seed=0, label=b"MiMCHash")
    a.transcript._label = b"MiMCHash"
    inst = a.prover.instance()
    assert (inst.n, a.gens_capacity, inst.q, inst.m) == (65124, 65536, 130250, 69)
    ctx.gens_ensure(65536)
    G, Hh = ctx.gens_export(0, 65536)
    og = O.Gens(compressed=(G, Hh))            # GPU generators (checked against the oracle's derivation in test_generators_*)
    assert og.export(65000, 100) == O.Gens(65100).export(65000, 100)
    check_against_oracle(ctx, a.prover, a.transcript, a.commitments, 65536, a.replay, ogens=og)


def test_cfg4_full_merkle_2_20_roundtrip(ctx):
    """BASELINE.json config 4 at full size: the reference's own 2^20 circuit (merkle_tree_gadget.rs:473-545). The oracle prover
    would need minutes, so the size-independent property is used: the GPU proof must be accepted by the oracle VERIFIER (one
    2N-term MSM on the CPU) run on the verifier-side assembly, and a tampered proof / wrong root must be rejected."""
    a = workloads.merkle_full_tree(ctx, leaves=512, seed=None, label=b"MerkleTree")
    inst = a.prover.instance()
    assert (inst.n, a.gens_capacity, inst.q, inst.m) == (993384, 1048576, 1986769, 512)
    state = a.transcript.state
    proof = a.prover.prove(bpg.BulletproofGens(ctx, a.gens_capacity), bytes(range(32)))
    assert len(proof) == 14 * 32 + (2 * 20 + 2) * 32 == 1792
    G, Hh = ctx.gens_export(0, a.gens_capacity)
    og = O.Gens(compressed=(G, Hh))
    sample = O.Gens(4096)
    assert og.export(0, 4096) == sample.export(0, 4096)
    tv = bpg.Transcript(b"MerkleTree")
    v = bpg.Verifier(tv)
    a.replay(v)
    vi = v.instance()
    assert tv.state == state and v.get_num_vars() == inst.n
    vc = to_oracle(vi)
    vstate = tv.state                                                    # Verifier::verify consumes the transcript: keep the pre-verify state
    assert O.verify(og, vstate, vc, b"".join(a.commitments), proof) == 0
    assert v.is_valid(proof, ctx, a.gens_capacity)                      # GPU verifier on the 2^20 circuit
    bad = bytearray(proof); bad[700] ^= 0x40
    tv2 = bpg.Transcript(b"MerkleTree"); v2 = bpg.Verifier(tv2); a.replay(v2)
    assert not v2.is_valid(bytes(bad), ctx, a.gens_capacity)
    assert O.verify(og, vstate, vc, b"".join(a.commitments), bytes(bad)) != 0
    # determinism: same seed -> same bytes; different seed -> different proof that still verifies
    res = ctx.upload(inst)
    p2, _ = res.prove(state, inst.v_blinding, bytes(range(32)))
    p3, _ = res.prove(state, inst.v_blinding, bytes(32))
    res.free()
    assert p2 == proof and p3 != proof
    assert O.verify(og, vstate, vc, b"".join(a.commitments), p3) == 0
    assert ctx.verify_flat(vi, vstate, b"".join(a.commitments), p3) == 0


def test_skewed_witness_distributions(ctx):
    """range-proof style witnesses (bits) and constant vectors put most MSM terms into a handful of buckets"""
    t = LabeledTranscript(b"skew")
    p = bpg.Prover(ctx, t)
    x = (2**200 - 12345).to_bytes(32, "little")
    cs, vs = p.commit_many([x], [rs(b"sk", 0)])
    bpg.range_proof(p, vs[0], 200, x)
    for _ in range(300):                       # 300 multipliers with identical assignments
        l, r, o = p.allocate_multiplier((sc(7), sc(9)))
        p.constrain(bpg.LinearCombination.of(o) - sc(63))
    check_against_oracle(ctx, p, t, cs, 512)


@pytest.mark.parametrize("tt_lg,orig_lg", [(4, 6), (8, 0), (3, 10), (12, 14)])
def test_tail_start_of_folded_and_of_original_generators(tt_lg, orig_lg, monkeypatch):
    """Two thresholds: a circuit of N <= 2^orig_lg freezes the ORIGINAL generators at round 0 (tables built once, A_I / A_O / S are table sums);
    a larger one folds down to 2^tt_lg and freezes the FOLDED generators there (tables per proof).  Any pair gives the oracle's bytes."""
    monkeypatch.setenv("BPG_TT_LG", str(tt_lg))
    monkeypatch.setenv("BPG_TT_ORIG_LG", str(orig_lg))
    c = bpg.Context(0)
    try:
        for a in (workloads.mimc_preimage(c, nbytes=20, seed=3, label=b"MiMCHash"), workloads.bounds_check_64(c, seed=9)):     # N = 1024 (n = 972) and N = 64
            inst = a.prover.instance()
            c.gens_ensure(a.gens_capacity)
            res = c.upload(inst)
            for seed in (bytes(range(32)), bytes(32)):          # the second proof reuses the tables of the original generators where they were built
                proof, st_after = res.prove(a.transcript.state, inst.v_blinding, seed, 0)
                rc, want, st_want = O.prove(O.Gens(a.gens_capacity), a.transcript.state, to_oracle(inst), inst.v_blinding, seed, O.FLAG_FAST_MSM)
                assert rc == 0 and proof == want and st_after == st_want, (tt_lg, orig_lg, inst.n)
            res.free()
    finally:
        c.close()


@pytest.mark.parametrize("env", [{}, {"BPG_FOLD_QUAD_W": "0"}, {"BPG_FOLD_ADAPT": "2"}, {"BPG_FOLD_ADAPT": "2", "BPG_FOLD_REG_W": "0"}])
@pytest.mark.parametrize("group", [1, 2, 3])
def test_small_folds_on_multiples_the_kernel_makes_itself(group, env, monkeypatch):
    """k_fold_points_quadw (a proof alone: four lanes per output) and k_fold_points_regw (BPG_FOLD_ADAPT=2, the shared-device variants: one lane per output)
    fold the groups after the first on width-4 NAF steps against 3P, 5P, 7P that the kernel makes and stores itself, one step list per class - with one,
    three and seven terms per output (groups of 1, 2, 3 rounds on N = 4096 and N = 16384: the folds whose outputs fill whole blocks take them, the smaller
    ones the register forms) - and with both switched off (plain NAF, addends in registers): the oracle's bytes every time."""
    monkeypatch.setenv("BPG_TT_LG", "0")
    monkeypatch.setenv("BPG_FOLD_GROUP", str(group))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = bpg.Context(0)
    try:
        for nbytes, cap in ((100, 4096), (500, 16384)):                                    # n = 3,888 and n = 15,552
            a = workloads.mimc_preimage(c, nbytes=nbytes, seed=32, label=b"MiMCHash")
            inst = a.prover.instance()
            assert a.gens_capacity == cap
            c.gens_ensure(cap)
            res = c.upload(inst)
            for seed in (bytes(range(32)), bytes([5]) * 32)[:2 if cap == 4096 else 1]:
                proof, st_after = res.prove(a.transcript.state, inst.v_blinding, seed, 0)
                rc, want, st_want = O.prove(O.Gens(cap), a.transcript.state, to_oracle(inst), inst.v_blinding, seed, O.FLAG_FAST_MSM)
                assert rc == 0 and proof == want and st_after == st_want, (group, env, cap)
            res.free()
    finally:
        c.close()


@pytest.mark.parametrize("tt_lg,group,split,wnaf,quad", [
    (0, 1, 1, 6, 1), (0, 2, 1, 6, 1), (0, 3, 1, 6, 1), (0, 5, 1, 6, 1), (3, 1, 1, 6, 1), (3, 2, 1, 6, 1), (3, 3, 1, 6, 1),
    (5, 4, 1, 6, 1), (8, 2, 1, 6, 1), (9, 5, 1, 6, 1), (11, 3, 1, 6, 1), (0, 2, 0, 0, 1), (0, 3, 0, 0, 1), (3, 4, 0, 0, 1),
    (0, 4, 1, 6, 1), (0, 2, 0, 6, 1), (0, 3, 0, 6, 1), (3, 4, 0, 5, 1), (0, 3, 0, 4, 1), (0, 1, 0, 3, 1), (0, 5, 0, 7, 1), (0, 3, 0, 8, 1), (0, 2, 1, 8, 1),
    (6, 3, 0, 6, 1), (0, 3, 1, 6, 0), (3, 2, 1, 6, 0), (0, 2, 1, 6, 0), (5, 4, 1, 6, 0), (0, 4, 1, 0, 0)])
def test_ipa_schedules_give_identical_proofs(tt_lg, group, split, wnaf, quad, monkeypatch):
    """The inner-product argument can be scheduled in many ways - generator folds grouped over `group` rounds, generators frozen
    below 2^tt_lg with window tables, or plain round-by-round folding (tt_lg = 0, group = 1).  Every schedule must give the
    oracle's bytes, including the first-round padding classes (n < N) falling on any term of a grouped fold, for each of the
    fold kernels (four lanes per output, 4-wave split, addends in registers, addends from memory for groups of 5) and for the width-w NAF fold of the
    first group against the odd multiples of the original generators (wnaf = 0: off; it replaces the register kernels there)."""
    monkeypatch.setenv("BPG_TT_LG", str(tt_lg))
    monkeypatch.setenv("BPG_FOLD_WNAF", str(wnaf))
    monkeypatch.setenv("BPG_FOLD_QUAD", str(quad))          # small folds: four lanes per output (1) or the four-wave split kernel (0)
    monkeypatch.setenv("BPG_FOLD_GROUP", str(group))
    if not split:
        monkeypatch.setenv("BPG_FOLD_SPLIT", "0")      # addends-in-registers fold kernels instead of the 4-wave latency variant
    c = bpg.Context(0)                       # the schedule is read when the context is created
    try:
        # 1-block MiMC preimage: n = 972, N = 1024 (52 padding generators)
        a = workloads.mimc_preimage(c, nbytes=20, seed=3, label=b"MiMCHash")
        inst = a.prover.instance()
        assert (inst.n, a.gens_capacity) == (972, 1024)
        og = O.Gens(1024)
        c.gens_ensure(1024)
        res = c.upload(inst)
        proof, st_after = res.prove(a.transcript.state, inst.v_blinding, bytes(range(32)), 0)
        rc, want, st_want = O.prove(og, a.transcript.state, to_oracle(inst), inst.v_blinding, bytes(range(32)), O.FLAG_FAST_MSM)
        assert rc == 0 and proof == want and st_after == st_want
        res.free()
        # n just above N/2 (worst padding: almost half of the generators are padding) and n = N
        for nbits in (33, 64):
            b = workloads.bounds_check_64(c, seed=nbits) if nbits == 64 else None
            if b is None:
                t = bpg.Transcript(b"RangeProof")
                p = bpg.Prover(c, t)
                x = bpg.be_to_scalar(bytes([1, 2, 3, 4, 1]))
                bpg.range_proof(p, x, 40, x)                               # 40 multipliers, N = 64: 24 padding generators
                inst2, state2, cap = p.instance(), t.state, 64
            else:
                inst2, state2, cap = b.prover.instance(), b.transcript.state, b.gens_capacity
            c.gens_ensure(cap)
            res = c.upload(inst2)
            proof, _ = res.prove(state2, inst2.v_blinding, bytes(32), 0)
            rc, want, _ = O.prove(O.Gens(cap), state2, to_oracle(inst2), inst2.v_blinding, bytes(32), O.FLAG_FAST_MSM)
            assert rc == 0 and proof == want, (tt_lg, group, nbits)
            res.free()
    finally:
        c.close()


@pytest.mark.parametrize("blocks,N", [(33, 1 << 15), (134, 1 << 17)])
def test_fold_group_sizes_above_the_tail(ctx, blocks, N):
    """Byte parity where the generator folds are live with the default schedule: N = 2^15 (one fold of a single round, then the
    table-driven tail) and N = 2^17 (one fold of three rounds with all seven addends in registers, padding generators on the
    last term)."""
    a = workloads.mimc_preimage(ctx, nbytes=32 * (blocks - 1) + 5, seed=blocks, label=b"MiMCHash")
    inst = a.prover.instance()
    assert inst.n == 972 * blocks and a.gens_capacity == N
    ctx.gens_ensure(N)
    G, Hh = ctx.gens_export(0, N)
    og = O.Gens(compressed=(G, Hh))
    res = ctx.upload(inst)
    proof, st_after = res.prove(a.transcript.state, inst.v_blinding, bytes(range(32)), 0)
    rc, want, st_want = O.prove(og, a.transcript.state, to_oracle(inst), inst.v_blinding, bytes(range(32)), O.FLAG_FAST_MSM)
    assert rc == 0 and proof == want and st_after == st_want
    res.free()


# (the fold profiles at N = 2^19 and 2^20 - serving, one-shot, two-part split, register kernels - are compared with oracle-generated fixtures in
# tests/test_proof_fixtures.py::test_gpu_reproduces_full_size_fixtures)


def test_expanded_blinding_dialect_matches_oracle_and_verifies(ctx):
    """BPG_FLAG_EXPANDED_BLINDING (opt-in, not upstream's derivation of s_L, s_R): same bytes as the oracle's independent
    restatement (SHAKE256 over one TranscriptRng draw), accepted by both verifiers, different from the default dialect."""
    for maker in (lambda: workloads.bounds_check_64(ctx, seed=5), lambda: workloads.mimc_preimage(ctx, nbytes=70, seed=9, label=b"MiMCHash")):
        a = maker()
        inst = a.prover.instance()
        state = a.transcript.state
        ctx.gens_ensure(a.gens_capacity)
        og = O.Gens(a.gens_capacity)
        res = ctx.upload(inst)
        base, _ = res.prove(state, inst.v_blinding, bytes(range(32)), 0)
        for flags in (bpg.FLAG_EXPANDED_BLINDING, bpg.FLAG_EXPANDED_BLINDING | bpg.FLAG_COMPACT_1PHASE):
            proof, st_after = res.prove(state, inst.v_blinding, bytes(range(32)), flags)
            rc, want, st_want = O.prove(og, state, to_oracle(inst), inst.v_blinding, bytes(range(32)), flags | O.FLAG_FAST_MSM)
            assert rc == 0 and proof == want and st_after == st_want
            enc = flags & 3                                        # the verifier only needs the encoding dialect
            coms = b"".join(a.commitments)
            assert O.verify(og, state, to_oracle(inst), coms, proof, flags=enc) == 0
            assert res.verify(state, coms, proof, flags=enc) == 0
            assert res.verify(state, coms, proof, flags=flags) == 0       # and ignores the prover-only bit
        assert proof != base
        res.free()


def _rand_scalar(rnd):
    kind = rnd.randrange(6)
    if kind == 0:
        return 0
    if kind == 1:
        return 1
    if kind == 2:
        return bpg.L - 1
    if kind == 3:
        return rnd.randrange(1 << 16)
    return rnd.randrange(bpg.L)


@pytest.mark.parametrize("seed", range(12))
def test_random_circuits_match_oracle(ctx, seed):
    """Randomly assembled satisfiable R1CS instances through the ConstraintSystem surface (commit, multiply, allocate_multiplier,
    constrain): every variable kind, repeated variables, zero / small / maximal coefficients and assignments, multipliers whose
    inputs are linear combinations, n anywhere between a power of two and the next.  GPU bytes == oracle bytes, both verifiers agree."""
    import random
    rnd = random.Random(1000 + seed)
    S = bpg.scalar_from_int
    t = LabeledTranscript(b"random-%d" % seed)
    p = bpg.Prover(ctx, t)
    values = {}                                       # Variable -> int assignment
    commitments = []
    m = rnd.randrange(0, 5)
    for _ in range(m):
        v = _rand_scalar(rnd)
        com, var = p.commit(S(v), S(rnd.randrange(bpg.L)))
        values[int(var)] = v; commitments.append(com)
    target_n = rnd.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 64, 100, 128, 129, 257])

    def rand_lc():
        terms, val = [], 0
        for _ in range(rnd.randrange(0, 4)):
            if values and rnd.random() < 0.8:
                var = rnd.choice(list(values))
                c = _rand_scalar(rnd)
                terms.append((bpg.Variable(var), S(c))); val = (val + c * values[var]) % bpg.L
            else:
                c = _rand_scalar(rnd)
                terms.append((bpg.Variable.One(), S(c))); val = (val + c) % bpg.L
        return bpg.LinearCombination(terms), val

    n = 0
    while n < target_n:
        if rnd.random() < 0.5:
            l, r = _rand_scalar(rnd), _rand_scalar(rnd)
            a, b, o = p.allocate_multiplier((S(l), S(r)))
        else:
            (ll, l), (rl, r) = rand_lc(), rand_lc()
            a, b, o = p.multiply(ll, rl)
        values[int(a)], values[int(b)], values[int(o)] = l, r, l * r % bpg.L
        n += 1
        for _ in range(rnd.randrange(0, 3)):          # constraints that hold: lc - value(lc) = 0
            lc, val = rand_lc()
            p.constrain(lc - bpg.LinearCombination([(bpg.Variable.One(), S(val))]))
    inst = p.instance()
    assert inst.n == target_n and inst.m == m
    cap = 1
    while cap < max(inst.n, 1):
        cap *= 2
    check_against_oracle(ctx, p, t, commitments, cap, flags_list=(0, rnd.choice([1, 2, 3])), seed=bytes([seed]) * 32)


def test_pool_batch_matches_individual_proofs(ctx):
    """bpg_pool_prove: a batch of independent instances (different circuits, sizes and dialects) on 4 worker contexts gives the bytes
    of proving them one by one; a bad item is reported without losing the others."""
    makers = [lambda: workloads.bounds_check_64(ctx, seed=1), lambda: workloads.mimc_preimage(ctx, nbytes=20, seed=2, label=b"MiMCHash"),
              lambda: workloads.bounds_check_64(ctx, seed=3), lambda: workloads.mimc_preimage(ctx, nbytes=70, seed=4, label=b"MiMCHash"),
              lambda: workloads.merkle_full_tree(ctx, leaves=2, seed=5), lambda: workloads.bounds_check_64(ctx, seed=6),
              lambda: workloads.mimc_preimage(ctx, nbytes=33, seed=7, label=b"MiMCHash")]
    items, want = [], []
    for k, mk in enumerate(makers):
        a = mk()
        inst = a.prover.instance()
        flags = k % 4
        seed = bytes([k + 1]) * 32
        ctx.gens_ensure(a.gens_capacity)
        res = ctx.upload(inst)
        want.append(res.prove(a.transcript.state, inst.v_blinding, seed, flags))
        res.free()
        items.append((inst, a.transcript.state, inst.v_blinding, seed, flags))
    pool = bpg.ProverPool(0, workers=4, gens_capacity=4096)
    try:
        got = pool.prove_batch(items)
        assert got == want
        assert pool.prove_batch(items[:2]) == want[:2]                 # fewer items than workers
        assert pool.prove_batch([]) == []
        bad = list(items)
        bad[3] = (items[3][0], items[3][1], items[3][2][:-32], items[3][3], items[3][4])     # one blinding factor short -> m mismatch
        inst3 = items[3][0]
        saved_m = inst3.m
        with pytest.raises(bpg.BpgError) as e:
            inst3.m = saved_m - 1 if saved_m else 0
            try:
                pool.prove_batch(bad)
            finally:
                inst3.m = saved_m
        assert "item 3" in str(e.value)
    finally:
        pool.close()


def test_speculative_blinding_stream_changes_no_byte(ctx):
    """bpg_prover_start_blinding / bpg_blinding_begin (extension): the TranscriptRng chain started before the constraints exist gives the
    same proof as the chain run inside prove(); a stream that no longer matches (another commitment, another seed, too small) is dropped."""
    seed, other = bytes(range(32)), bytes(range(1, 33))
    x = bpg.be_to_scalar(bytes.fromhex("0522a64d7b931e"))
    leaf = bpg.be_to_scalar(bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749"))
    pat = "(((I I) (I I)) ((I I) (I I)))"
    probe = bpg.Prover(None, bpg.Transcript(b"probe"))
    bpg.MerkleTree256(bytes(32), [leaf] * 8, [], pat).prove(probe, [], [])
    root = probe.instance().aO[-32:]

    def build(early, extra_commit=False, prove_seed=seed, max_mult=1 << 14, flags=0):
        t = bpg.Transcript(b"stream")
        p = bpg.Prover(ctx, t)
        com, var = p.commit(x, bpg.be_to_scalar(b"\x07" * 31))
        if early:
            p.start_blinding(seed, max_mult)
        if extra_commit:
            p.commit(leaf, bpg.be_to_scalar(b"\x09" * 31))
        bpg.range_proof(p, var, 56, x)
        bpg.MerkleTree256(root, [leaf] * 8, [], pat).prove(p, [], [])          # n = 13,608 + 56: the stream crosses several snapshots
        inst, state = p.instance(), t.state
        return p.prove(bpg.BulletproofGens(ctx, 1 << 14), prove_seed, flags), inst, state

    og = O.Gens(1 << 14)
    plain, inst, state = build(False)
    rc, want, _ = O.prove(og, state, to_oracle(inst), inst.v_blinding, seed, O.FLAG_FAST_MSM)
    assert rc == 0 and plain == want
    assert build(True)[0] == plain                                   # stream used
    assert build(True, max_mult=1000)[0] == plain                    # too small for this circuit: dropped
    assert build(True, flags=bpg.FLAG_EXPANDED_BLINDING)[0] == build(False, flags=bpg.FLAG_EXPANDED_BLINDING)[0]
    p2, i2, s2 = build(True, prove_seed=other)                       # proved under another seed: dropped
    rc, want2, _ = O.prove(og, s2, to_oracle(i2), i2.v_blinding, other, O.FLAG_FAST_MSM)
    assert rc == 0 and p2 == want2 and p2 != plain
    p3, i3, s3 = build(True, extra_commit=True)                      # a commitment after the start: dropped
    rc, want3, _ = O.prove(og, s3, to_oracle(i3), i3.v_blinding, seed, O.FLAG_FAST_MSM)
    assert rc == 0 and p3 == want3
    # context-level entry point on resident circuits whose 2n sits on, just past and far below a snapshot boundary (4096 draws)
    one, zero = (1).to_bytes(32, "little"), bytes(32)
    for n in (2048, 2049, 1):
        t = bpg.Transcript(b"stream2")
        p = bpg.Prover(ctx, t)
        for i in range(n):
            p.allocate_multiplier((one, zero) if i & 1 else (zero, one))
        inst, state = p.instance(), t.state
        cap = 1
        while cap < n:
            cap *= 2
        ctx.gens_ensure(cap)
        res = ctx.upload(inst)
        a, _ = res.prove(state, b"", seed, 0)
        ctx.blinding_begin(state, b"", seed, cap)
        b, _ = res.prove(state, b"", seed, 0)
        c, _ = res.prove(state, b"", seed, 0)                        # stream consumed: plain path again
        assert a == b == c
        rc, want, _ = O.prove(O.Gens(cap), state, to_oracle(inst), b"", seed, O.FLAG_FAST_MSM)
        assert rc == 0 and a == want


def test_chain_knobs_from_the_environment_change_no_byte(monkeypatch):
    """BPG_CHAIN_WORKERS / BPG_CHAIN_LANES / BPG_SYNC_BLOCKING (the environment fall-backs of bpg_config.chain_workers / chain_lanes / blocking_sync, read
    when the context is created): two chain threads with two lockstep lanes each and blocking stream waits draw the queued chains of a sequence of
    proofs; every proof equals the one made without any stream."""
    monkeypatch.setenv("BPG_CHAIN_WORKERS", "2")
    monkeypatch.setenv("BPG_CHAIN_LANES", "2")
    monkeypatch.setenv("BPG_SYNC_BLOCKING", "1")
    c = bpg.Context(0)
    plain = bpg.Context(0, chain_workers=1, chain_lanes=1, blocking_sync=False)
    try:
        a = workloads.mimc_preimage(c, nbytes=300, seed=9, label=b"MiMCHash")        # 10 blocks: n = 9,720, N = 2^14: the draws cross two snapshots
        inst, state = a.prover.instance(), a.transcript.state
        for x in (c, plain):
            x.gens_ensure(a.gens_capacity)
        res, ref = c.upload(inst), plain.upload(inst)
        seeds = [bytes([k + 3]) * 32 for k in range(5)]
        want = [ref.prove(state, inst.v_blinding, s, 0)[0] for s in seeds]
        for s in seeds:                                              # workers * lanes + 1 = 5 streams may be alive
            c.blinding_begin(state, inst.v_blinding, s, inst.n)
        got = [res.prove(state, inst.v_blinding, s, 0)[0] for s in seeds]
        assert got == want
        rc, oracle, _ = O.prove(O.Gens(a.gens_capacity), state, to_oracle(inst), inst.v_blinding, seeds[0], O.FLAG_FAST_MSM)
        assert rc == 0 and got[0] == oracle
        res.free(); ref.free()
    finally:
        c.close(); plain.close()


def test_queued_blinding_streams_serve_a_sequence_of_proofs(ctx):
    """The chain worker draws queued streams one at a time, in order; up to two stay alive (one proving stream's sequence: begin(i+1), prove(i)).
    Every proof of the sequence equals the stand-alone proof of the same seed; a third begin retires the oldest stream; proofs whose
    stream was retired, never queued or queued out of order still come out byte-identical."""
    a = workloads.mimc_preimage(ctx, nbytes=200, seed=3)               # n = 6,804: 2n crosses three snapshots of 4,096 draws
    inst, state = a.prover.instance(), a.transcript.state
    ctx.gens_ensure(a.gens_capacity)
    res = ctx.upload(inst)
    seeds = [bytes([k + 1]) * 32 for k in range(6)]
    alone = [res.prove(state, inst.v_blinding, s, 0)[0] for s in seeds]
    rc, want, _ = O.prove(O.Gens(a.gens_capacity), state, to_oracle(inst), inst.v_blinding, seeds[0], O.FLAG_FAST_MSM)
    assert rc == 0 and alone[0] == want
    assert len(set(alone)) == len(alone)
    # the sequence of bench.py
    ctx.blinding_begin(state, inst.v_blinding, seeds[0], inst.n)
    got = []
    for k, s in enumerate(seeds):
        if k + 1 < len(seeds):
            ctx.blinding_begin(state, inst.v_blinding, seeds[k + 1], inst.n)
        got.append(res.prove(state, inst.v_blinding, s, 0)[0])
    assert got == alone
    assert ctx.chain_cpu() >= 0
    # three begins in a row: the first stream gives way; proving in another order than queued, and a seed that was never queued
    for s in seeds[:3]:
        ctx.blinding_begin(state, inst.v_blinding, s, inst.n)
    assert res.prove(state, inst.v_blinding, seeds[2], 0)[0] == alone[2]
    assert res.prove(state, inst.v_blinding, seeds[0], 0)[0] == alone[0]      # retired: drawn inside prove
    assert res.prove(state, inst.v_blinding, seeds[4], 0)[0] == alone[4]      # never queued
    assert res.prove(state, inst.v_blinding, seeds[1], 0)[0] == alone[1]      # still alive, consumed now
    # a stream sized for fewer multipliers than the circuit has is ignored; a context with queued streams can be torn down
    ctx.blinding_begin(state, inst.v_blinding, seeds[5], 100)
    assert res.prove(state, inst.v_blinding, seeds[5], 0)[0] == alone[5]
    c2 = bpg.Context(0)
    c2.blinding_begin(state, inst.v_blinding, seeds[0], 1 << 16)
    c2.blinding_begin(state, inst.v_blinding, seeds[1], 1 << 16)
    c2.close()
    # several chain threads (bpg_ctx_set_chain_workers): workers + 1 streams alive, drawn side by side; bench.py's deep sequence
    ctx.set_chain_workers(3)
    try:
        queued, got = 0, []
        for k, s in enumerate(seeds):
            while queued < len(seeds) and queued <= k + 3:
                ctx.blinding_begin(state, inst.v_blinding, seeds[queued], inst.n)
                queued += 1
            got.append(res.prove(state, inst.v_blinding, s, 0)[0])
        assert got == alone
        for s in seeds[:5]:                                      # a fifth begin retires the oldest of four
            ctx.blinding_begin(state, inst.v_blinding, s, inst.n)
        assert [res.prove(state, inst.v_blinding, s, 0)[0] for s in (seeds[4], seeds[0], seeds[2])] == [alone[4], alone[0], alone[2]]
    finally:
        ctx.set_chain_workers(1)                                 # drops what is still queued
    assert res.prove(state, inst.v_blinding, seeds[3], 0)[0] == alone[3]
    # one chain thread, several streams in lockstep (bpg_ctx_set_chain_lanes: the sponges of up to eight queued streams in the lanes of ZMM
    # registers): streams join and leave at snapshot boundaries; circuits of different sizes share a thread; same bytes throughout
    b2 = workloads.mimc_preimage(ctx, nbytes=500, seed=9)              # another n: its streams run longer than the others'
    inst2, state2 = b2.prover.instance(), b2.transcript.state
    ctx.gens_ensure(b2.gens_capacity)
    res2 = ctx.upload(inst2)
    alone2 = [res2.prove(state2, inst2.v_blinding, s, 0)[0] for s in seeds[:2]]
    for lanes in (2, 5, 8):
        ctx.set_chain_lanes(lanes)
        try:
            queued, got = 0, []
            for k, s in enumerate(seeds):
                while queued < len(seeds) and queued <= k + lanes - 1:
                    ctx.blinding_begin(state, inst.v_blinding, seeds[queued], inst.n)
                    queued += 1
                got.append(res.prove(state, inst.v_blinding, s, 0)[0])
            assert got == alone, lanes
            ctx.blinding_begin(state2, inst2.v_blinding, seeds[0], inst2.n)      # mixed sizes in one lockstep group
            ctx.blinding_begin(state, inst.v_blinding, seeds[1], inst.n)
            ctx.blinding_begin(state2, inst2.v_blinding, seeds[1], inst2.n)
            assert res.prove(state, inst.v_blinding, seeds[1], 0)[0] == alone[1]
            assert res2.prove(state2, inst2.v_blinding, seeds[1], 0)[0] == alone2[1]
            assert res2.prove(state2, inst2.v_blinding, seeds[0], 0)[0] == alone2[0]
        finally:
            ctx.set_chain_lanes(1)
    assert res.prove(state, inst.v_blinding, seeds[2], 0)[0] == alone[2]
    res2.free()
    res.free()


def test_chain_pool_serves_several_contexts(ctx):
    """bpg_chain_pool_*: ONE set of chain threads (here one single-lane thread and one that draws three chains in lockstep) draws the blinding
    streams of every attached context - three contexts proving on their own host threads at once, circuits of two sizes, each keeping the chain of
    its next proof queued.  Same bytes as stand-alone proofs; a third begin on a context retires its oldest stream; detaching (pool = None, or
    bpg_ctx_set_chain_workers) and destroying contexts while the pool goes on serving the others changes nothing; a pool destroyed under an attached
    context sends that context back to its own chain worker."""
    import threading
    small = workloads.mimc_preimage(ctx, nbytes=200, seed=21)            # n = 6,804
    big = workloads.mimc_preimage(ctx, nbytes=900, seed=22)              # n = 28,188
    seeds = [bytes([40 + k]) * 32 for k in range(4)]
    want = {}
    for tag, a in (("small", small), ("big", big)):
        inst = a.prover.instance()
        ctx.gens_ensure(a.gens_capacity)
        r = ctx.upload(inst)
        want[tag] = [r.prove(a.transcript.state, inst.v_blinding, s, 0)[0] for s in seeds]
        r.free()
    pool = bpg.ChainPool([1, 3])
    assert pool.capacity == 4
    ctxs = [bpg.Context(0) for _ in range(3)]
    errs, got = [], {}
    # every thread gets its own copy of the flattened instance, taken here: a prover (like a context) is used by one host thread at a time
    jobs = [(("small", small), ("big", big))[k % 2] for k in range(3)]
    jobs = [(tag, a, a.prover.instance(), a.transcript.state) for tag, a in jobs]

    def work(k):
        try:
            c = ctxs[k]
            tag, a, inst, state = jobs[k]
            c.gens_ensure(a.gens_capacity)
            r = c.upload(inst)
            c.attach_chain_pool(pool, 2)
            out, queued = [], 0
            for i, s in enumerate(seeds):
                while queued < len(seeds) and queued <= i + 1:           # the chain of the next proof is queued before this one is proved
                    c.blinding_begin(state, inst.v_blinding, seeds[queued], inst.n)
                    queued += 1
                out.append(r.prove(state, inst.v_blinding, s, 0)[0])
            # three begins with two streams allowed: the oldest is retired, its proof draws inside the call; the others are served by the pool
            for s in seeds[:3]:
                c.blinding_begin(state, inst.v_blinding, s, inst.n)
            out += [r.prove(state, inst.v_blinding, s, 0)[0] for s in (seeds[2], seeds[0], seeds[1])]
            if k == 0:
                c.attach_chain_pool(None)                                # back to its own worker
            elif k == 1:
                c.set_chain_workers(2)                                   # detaches as well
            c.blinding_begin(state, inst.v_blinding, seeds[3], inst.n)
            out.append(r.prove(state, inst.v_blinding, seeds[3], 0)[0])
            r.free()
            got[k] = (tag, out)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(3):
        tag, out = got[k]
        assert out == want[tag] + [want[tag][2], want[tag][0], want[tag][1], want[tag][3]], (k, tag)
    ctxs[2].close()                                                      # destroyed while attached: the pool goes on
    inst = small.prover.instance()
    r = ctxs[0].upload(inst)
    ctxs[0].attach_chain_pool(pool, 2)
    ctxs[0].blinding_begin(small.transcript.state, inst.v_blinding, seeds[1], inst.n)
    assert r.prove(small.transcript.state, inst.v_blinding, seeds[1], 0)[0] == want["small"][1]
    # the pool is destroyed while a context is still attached (a host's mistake): the context notices and goes back to its own chain worker
    ctxs[1].attach_chain_pool(pool, 2)
    pool.close()
    ctxs[1].gens_ensure(small.gens_capacity)
    r1 = ctxs[1].upload(inst)
    ctxs[1].blinding_begin(small.transcript.state, inst.v_blinding, seeds[2], inst.n)
    assert r1.prove(small.transcript.state, inst.v_blinding, seeds[2], 0)[0] == want["small"][2]
    ctxs[0].blinding_begin(small.transcript.state, inst.v_blinding, seeds[3], inst.n)
    assert r.prove(small.transcript.state, inst.v_blinding, seeds[3], 0)[0] == want["small"][3]
    r.free(); r1.free()
    for c in ctxs[:2]:
        c.close()


@pytest.mark.parametrize("parts,wnaf,group,budget_gb", [(1, 8, 3, None), (2, 8, 3, None), (4, 8, 3, None), (2, 5, 2, None), (4, 6, 1, None), (4, 3, 5, None), (1, 6, 4, None),
                                                         (4, 8, 3, "0.01"), (4, 8, 2, "0.03"), (2, 7, 3, "0.0001")])
def test_split_scalar_fold_gives_identical_proofs(parts, wnaf, group, budget_gb, monkeypatch):
    """The first generator fold cuts its shared scalars into `parts` pieces on tables of 2^(j*L) * P (one chain of L = ceil(254/parts) doublings
    instead of 253): same bytes as the oracle for every split, NAF width and group size, with padding generators in the first group; a table
    budget (BPG_FOLD_TABLE_GB) below what the requested profile needs makes the engine fall back to fewer parts, then narrower windows."""
    if budget_gb is not None:
        monkeypatch.setenv("BPG_FOLD_TABLE_GB", budget_gb)
    monkeypatch.setenv("BPG_TT_LG", "0")
    monkeypatch.setenv("BPG_FOLD_SPLIT", "0")            # the width-w NAF fold at this small size
    monkeypatch.setenv("BPG_FOLD_GROUP", str(group))
    monkeypatch.setenv("BPG_FOLD_WNAF", str(wnaf))
    monkeypatch.setenv("BPG_FOLD_PARTS", str(parts))
    monkeypatch.setenv("BPG_GENS_SHARE", "0")
    c = bpg.Context(0)
    try:
        for nbytes, seed in ((20, 3), (50, 5)):            # n = 972 (N = 1024: 52 padding generators) and n = 1944 (N = 2048)
            a = workloads.mimc_preimage(c, nbytes=nbytes, seed=seed, label=b"MiMCHash")
            inst = a.prover.instance()
            c.gens_ensure(a.gens_capacity)
            res = c.upload(inst)
            proof, st_after = res.prove(a.transcript.state, inst.v_blinding, bytes(range(32)), 0)
            rc, want, st_want = O.prove(O.Gens(a.gens_capacity), a.transcript.state, to_oracle(inst), inst.v_blinding, bytes(range(32)), O.FLAG_FAST_MSM)
            assert rc == 0 and proof == want and st_after == st_want
            res.free()
    finally:
        c.close()


@pytest.mark.parametrize("wide_gb", ["0", "24"])
def test_tail_on_original_generators_with_and_without_wide_tables(wide_gb, monkeypatch):
    """Circuits up to 2^14 multipliers freeze their generators at round 0; the tail then reads 8-bit window tables of the original generators
    (k_tt_round8, built once per device) unless BPG_TT_WIDE_GB forbids them: same bytes as the oracle either way, padding generators included,
    also for the second proof on the same context (cached tables) and for a circuit of another size on the same context."""
    monkeypatch.setenv("BPG_TT_WIDE_GB", wide_gb)
    monkeypatch.setenv("BPG_GENS_SHARE", "0")
    c = bpg.Context(0, profile="serving")                   # the one-shot default never builds them (4.3 GB at N = 4096)
    try:
        cases = [workloads.mimc_preimage(c, nbytes=20, seed=3, label=b"MiMCHash"),      # n = 972, N = 1024
                 workloads.bounds_check_64(c, seed=1),                                  # n = N = 128
                 workloads.mimc_preimage(c, nbytes=100, seed=4)]                        # N = 4096
        for a in cases:
            inst = a.prover.instance()
            c.gens_ensure(a.gens_capacity)
            res = c.upload(inst)
            og = O.Gens(a.gens_capacity)
            for k in range(2):
                proof, st_after = res.prove(a.transcript.state, inst.v_blinding, bytes([k + 7]) * 32, 0)
                rc, want, st_want = O.prove(og, a.transcript.state, to_oracle(inst), inst.v_blinding, bytes([k + 7]) * 32, O.FLAG_FAST_MSM)
                assert rc == 0 and proof == want and st_after == st_want
            res.free()
    finally:
        c.close()


def test_profiles_and_the_cumulative_table_budget(ctx, monkeypatch):
    """bpg_ctx_create = the one-shot profile (15 odd multiples per generator); bpg_ctx_create_ex chooses: serving = 255 multiples; a table budget counts
    every table the process holds on the device (both capacities below together), and what does not fit is replaced by the next smaller set, in the
    end by the register kernels.  Same bytes as the oracle every time."""
    monkeypatch.setenv("BPG_TT_LG", "0")
    monkeypatch.setenv("BPG_FOLD_SPLIT", "0")            # the width-w NAF fold at this small size
    monkeypatch.setenv("BPG_GENS_SHARE", "0")
    for k in ("BPG_PROFILE", "BPG_FOLD_WNAF", "BPG_FOLD_PARTS", "BPG_TABLE_GB", "BPG_FOLD_TABLE_GB"):
        monkeypatch.delenv(k, raising=False)
    row = lambda cap: 2 * cap * 96                          # one table of multiples: [G | H] in affine Niels form
    def run(c, nbytes, seed):
        a = workloads.mimc_preimage(c, nbytes=nbytes, seed=seed, label=b"MiMCHash")
        inst = a.prover.instance()
        c.gens_ensure(a.gens_capacity)
        res = c.upload(inst)
        proof, _ = res.prove(a.transcript.state, inst.v_blinding, bytes(range(32)), 0)
        rc, want, _ = O.prove(O.Gens(a.gens_capacity), a.transcript.state, to_oracle(inst), inst.v_blinding, bytes(range(32)), O.FLAG_FAST_MSM)
        assert rc == 0 and proof == want
        res.free()
        return a.gens_capacity
    for kw, tables in (({}, 15), ({"profile": "oneshot"}, 15), ({"profile": "serving"}, 255)):
        c = bpg.Context(0, **kw)
        try:
            before = c.table_bytes()
            cap = run(c, 20, 3)                            # N = 1024
            assert c.table_bytes() - before == tables * row(cap), (kw, c.table_bytes() - before)
            run(c, 20, 3)                                   # second proof: nothing new
            assert c.table_bytes() - before == tables * row(cap)
        finally:
            c.close()
        assert ctx.table_bytes() == before                  # freed with the last context of the generation
    # 5.5 MB on top of what the process already holds: the serving set of N = 1024 (255 x 196,608 B = 50 MB) shrinks - two parts, one part, width 7
    # (31 tables, 6.1 MB) - to width 6 unsplit (15 tables, 2.9 MB); the next capacity on the same device (N = 2048: 15 tables = 5.9 MB, 7 = 2.8 MB)
    # no longer fits beside it until width 4 (3 tables, 1.2 MB)
    before = ctx.table_bytes()
    c = bpg.Context(0, profile="serving", table_budget_gb=(before + 5.5e6) / 2**30)
    try:
        cap = run(c, 20, 3)
        assert c.table_bytes() - before == 15 * row(cap)
        c2 = bpg.Context(0, profile="serving", table_budget_gb=(before + 5.5e6) / 2**30)
        try:
            cap2 = run(c2, 50, 5)                          # N = 2048
            assert c2.table_bytes() - before == 15 * row(cap) + 3 * row(cap2)
        finally:
            c2.close()
    finally:
        c.close()
    # no budget at all: the register kernels fold, no table is built
    c = bpg.Context(0, table_budget_gb=1e-6)
    try:
        run(c, 20, 3)
        assert c.table_bytes() == before
    finally:
        c.close()
    with pytest.raises(bpg.BpgError):
        bpg.Context(0, profile=2, chain_lanes=9)


def test_failed_upload_of_blinding_draws_is_refused(ctx):
    """The chain worker uploads its draws into a device slab that is reused from proof to proof; an upload that fails must stop the proof
    (BPG_ERR_DEVICE), never let s_L, s_R be built from what the slab held before.  The next proof on the context is unaffected."""
    import ctypes as C
    a = workloads.mimc_preimage(ctx, nbytes=100, seed=11, label=b"MiMCHash")
    inst, state = a.prover.instance(), a.transcript.state
    ctx.gens_ensure(a.gens_capacity)
    res = ctx.upload(inst)
    seed = bytes([9]) * 32
    good, _ = res.prove(state, inst.v_blinding, seed, 0)
    ctx.blinding_begin(state, inst.v_blinding, seed, inst.n)
    assert res.prove(state, inst.v_blinding, seed, 0)[0] == good          # the stream path, uploads intact
    assert bpg.lib().bpg_test_fail_next_upload(ctx._h) == 0
    ctx.blinding_begin(state, inst.v_blinding, seed, inst.n)
    with pytest.raises(bpg.BpgError) as e:
        res.prove(state, inst.v_blinding, seed, 0)
    assert e.value.status == 7 and "upload" in str(e.value)
    assert res.prove(state, inst.v_blinding, seed, 0)[0] == good          # drawn inside the call
    ctx.blinding_begin(state, inst.v_blinding, seed, inst.n)
    assert res.prove(state, inst.v_blinding, seed, 0)[0] == good
    # a copy that is dropped WITHOUT an error: the slab still holds the marks it got when it changed owner; the conversion kernel sees them and the
    # proof is withheld (never built from what the slab held before)
    assert bpg.lib().bpg_test_drop_next_upload(ctx._h) == 0
    ctx.blinding_begin(state, inst.v_blinding, seed, inst.n)
    with pytest.raises(bpg.BpgError) as e:
        res.prove(state, inst.v_blinding, seed, 0)
    assert e.value.status == 7 and "stale" in str(e.value)
    ctx.blinding_begin(state, inst.v_blinding, seed, inst.n)
    assert res.prove(state, inst.v_blinding, seed, 0)[0] == good          # and the context goes on
    res.free()


def test_contexts_of_one_device_share_generator_tables(monkeypatch):
    """Contexts of one device share their generator tables and the odd multiples of the width-w NAF fold (engine.hip SharedTables): two contexts prove
    side by side on their own streams from the same tables, the tables outlive the context that derived them, and a context that asks for a
    larger capacity moves to a new generation without disturbing the others.  Same bytes as the oracle throughout."""
    import threading
    monkeypatch.setenv("BPG_TT_LG", "0")
    monkeypatch.setenv("BPG_FOLD_SPLIT", "0")            # the width-w NAF fold at this small size
    monkeypatch.setenv("BPG_FOLD_GROUP", "3")
    c1, c2 = bpg.Context(0), bpg.Context(0)
    a = workloads.mimc_preimage(c1, nbytes=20, seed=3, label=b"MiMCHash")
    inst = a.prover.instance()
    og = O.Gens(1024)
    seeds = [bytes([k]) * 32 for k in range(4)]
    want = []
    for s in seeds:
        rc, w, _ = O.prove(og, a.transcript.state, to_oracle(inst), inst.v_blinding, s, O.FLAG_FAST_MSM)
        assert rc == 0
        want.append(w)
    c1.gens_ensure(1024); c2.gens_ensure(1024)
    assert c1.gens_export(0, 8) == c2.gens_export(0, 8)
    r1, r2 = c1.upload(inst), c2.upload(inst)
    got, errs = {}, []

    def run(tag, r):
        try:
            got[tag] = [r.prove(a.transcript.state, inst.v_blinding, s, 0)[0] for s in seeds]
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=run, args=(1, r1)), threading.Thread(target=run, args=(2, r2))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs and got[1] == want and got[2] == want
    r1.free(); c1.close()                                  # the deriving context goes away; c2 keeps the tables alive
    assert r2.prove(a.transcript.state, inst.v_blinding, seeds[1], 0)[0] == want[1]
    c3 = bpg.Context(0)
    c3.gens_ensure(4096)                                   # a new generation beside the one c2 uses
    assert c3.gens_export(0, 1024) == c2.gens_export(0, 1024)      # the chains are prefixes of one another
    c2.gens_ensure(4096)                                   # c2 moves over (adopts c3's tables); its resident circuit still proves
    assert r2.prove(a.transcript.state, inst.v_blinding, seeds[2], 0)[0] == want[2]
    r2.free(); c2.close(); c3.close()


def test_generator_cache_on_disk(tmp_path, monkeypatch):
    """BPG_GENS_CACHE_DIR: the table written by one context is what a second one loads (same exported generators, same
    proof bytes); a file with a flipped byte (checksum), a truncated file, a wrong capacity in the header, and a file whose FIRST point was
    replaced with a consistent checksum (sample comparison against freshly derived generators) are all ignored and the table is re-derived."""
    import struct
    monkeypatch.setenv("BPG_GENS_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("BPG_GENS_SHARE", "0")              # a live context with tables of this capacity would otherwise be adopted before the file is looked at
    cap = 4096
    path = tmp_path / ("gens_%d.bpg" % cap)

    def table():
        c = bpg.Context(0)
        c.gens_ensure(cap)
        g, h = c.gens_export(0, cap)
        a = workloads.mimc_preimage(c, nbytes=100, seed=1)
        inst = a.prover.instance()
        proof, _ = c.upload(inst).prove(a.transcript.state, inst.v_blinding, bytes(range(32)), 0)
        c.close()
        return g + h, proof
    want, proof = table()                                  # derives, writes the file
    assert path.exists() and path.stat().st_size == 40 + 2 * cap * 96
    good = path.read_bytes()
    assert table() == (want, proof)                        # loads it

    def checksum(body):
        M = (1 << 64) - 1
        h = [0x9e3779b97f4a7c15, 0xc2b2ae3d27d4eb4f, 0x165667b19e3779f9, 0x27d4eb2f165667c5]
        words = struct.unpack("<%dQ" % (len(body) // 8), body)
        for i in range(0, len(words) - 3, 4):
            for k in range(4):
                x = ((h[k] ^ words[i + k]) * 0x100000001b3) & M
                h[k] = ((x << 29) | (x >> 35)) & M
        return (h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7) ^ len(body)) & M
    assert struct.unpack("<Q", good[32:40])[0] == checksum(good[40:])
    flipped = bytearray(good); flipped[40 + 96 * 1000 + 5] ^= 1
    swapped = bytearray(good); swapped[40:40 + 96] = good[40 + 96:40 + 192]          # G_0 := G_1, checksum made consistent: only the sample check sees it
    swapped[32:40] = struct.pack("<Q", checksum(bytes(swapped[40:])))
    wrongcap = bytearray(good); wrongcap[16:24] = struct.pack("<Q", 2 * cap)
    for bad in (bytes(flipped), good[:-7], bytes(swapped), bytes(wrongcap), b"", good + b"\x00"):
        path.write_bytes(bad)
        assert table() == (want, proof)
        assert path.read_bytes() == good                   # the re-derived table replaced the bad file
    # the cache trusts the file system, so it only touches what belongs to this user alone: a directory or file that others may write,
    # or a symbolic link in the file's place, is neither read nor written - the table is derived
    import os, stat
    assert stat.S_IMODE(path.stat().st_mode) & 0o022 == 0 and not list(tmp_path.glob("*.tmp.*"))
    os.chmod(path, 0o666)
    path.write_bytes(bytes(swapped))                       # would be refused by the sample check anyway; must not even be replaced now
    assert table() == (want, proof) and path.read_bytes() == bytes(swapped)
    os.chmod(path, 0o600); path.write_bytes(good)
    os.chmod(tmp_path, 0o777)
    path.unlink()
    assert table() == (want, proof) and not path.exists()  # a directory others can write: no cache file appears
    os.chmod(tmp_path, 0o700)
    real = tmp_path / "elsewhere.bin"; real.write_bytes(good)
    path.symlink_to(real)
    assert table() == (want, proof) and path.is_symlink() and real.read_bytes() == good
