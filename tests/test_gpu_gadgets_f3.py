"""Equality / Inequality / LessThan / SetMembership: the reference's positive and negative unit tests
(src/equality/equality_gadget.rs:50-197, inequality_gadget.rs:115-419, less_than_gadget.rs:88-333, set_membership_gadget.rs:134-403)
restated: prove on the GPU, byte-compare with the oracle, verify with the oracle AND the GPU verifier; unsatisfied statements
must be rejected by both."""
import hashlib
import pytest
import bulletproofs_gadgets_amd as bpg
import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
sc = lambda x: (x % R.L).to_bytes(32, "little")
rs = lambda tag, i: sc(int.from_bytes(hashlib.sha512(b"%s%d" % (tag, i)).digest(), "little"))
H = bytes.fromhex


@pytest.fixture(scope="module")
def ctx():
    return bpg.Context(0)


def to_oracle(inst):
    return O.FlatCircuit(inst.n, inst.m, inst.aL or None, inst.aR or None, inst.aO or None, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)


def run(ctx, label, build_prover, build_verifier, capacity, expect_ok):
    tp = bpg.Transcript(label)
    p = bpg.Prover(ctx, tp)
    coms = build_prover(p)
    inst, state = p.instance(), tp.state
    proof = p.prove(bpg.BulletproofGens(ctx, capacity), bytes(range(32)))
    og = O.Gens(capacity)
    rc, want, _ = O.prove(og, state, to_oracle(inst), inst.v_blinding, bytes(range(32)), O.FLAG_FAST_MSM)
    assert rc == 0 and proof == want
    assert O.satisfied(to_oracle(inst), inst.v) == expect_ok
    tv = bpg.Transcript(label)
    v = bpg.Verifier(tv)
    build_verifier(v, coms)
    vi = v.instance()
    assert tv.state == state and vi.commitments == b"".join(coms)
    assert (O.verify(og, tv.state, to_oracle(vi), vi.commitments, proof) == 0) == expect_ok
    assert v.is_valid(proof, ctx, capacity) == expect_ok


W_A = H("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")
LONG = bytes(range(1, 71))          # 70 bytes -> 3 scalars


@pytest.mark.parametrize("left,right,ok", [(W_A, W_A, True), (W_A, W_A[:-1] + b"\x48", False), (LONG, LONG, True),
                                           (LONG, LONG[:40] + b"\xff" + LONG[41:], False), (LONG, LONG[:33], False),
                                           # src/equality/equality_gadget.rs:52-197 (test_equality_gadget_1..4), (left, right, verdict); 1 = the first case above
                                           (H("0522a64d7b931e21760cf95aa15fcc793e8a52b42a56ab03afddec8beb668749"), W_A, False),
                                           (H("0522a64d7b931e21"), W_A, False),
                                           (H("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb6687493e8a52032a56ab03afddec8beb668749"),
                                            H("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb6687493e8a52032a56ab03afddec8beb668749"), True)])
def test_equality_instance_right_hand(ctx, left, right, ok):
    def bp(p):
        self_ = {}
        _, coms, vars_ = bpg.commit(p, left, [rs(b"eq", i) for i in range(4)])
        bpg.Equality(bpg.be_to_scalars(right)).prove(p, vars_, [])
        return coms

    def bv(v, coms):
        bpg.Equality(bpg.be_to_scalars(right)).verify(v, bpg.verifier_commit(v, coms), [])
    run(ctx, b"Equality", bp, bv, 1, ok)


def test_equality_witness_right_hand(ctx):
    def bp(p):
        _, c1, v1 = bpg.commit(p, LONG, [rs(b"e1", i) for i in range(3)])
        _, c2, v2 = bpg.commit(p, LONG, [rs(b"e2", i) for i in range(3)])
        bpg.Equality(v2).prove(p, v1, [])
        return c1 + c2

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        bpg.Equality(vs[3:]).verify(v, vs[:3], [])
    run(ctx, b"Equality", bp, bv, 1, True)


@pytest.mark.parametrize("left,right,ok", [(W_A, W_A[:-1] + b"\x48", True), (W_A, W_A, False), (LONG, LONG[:69] + b"\x00", True), (LONG, LONG, False),
                                           (b"\x05", b"\x07", True), (b"\x07", b"\x05", True)] + [
    # src/inequality/inequality_gadget.rs:126-419 (test_inequality_gadget_1..7): (left_assignment, right, verdict)
    (H("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749" * 2 + "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8ceb668749"),
     H("0522a64d7b931e21760cf955a15fcc733e8a52b42a56ab03afddec8beb668749" "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab02afddec8beb668749"
       "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8ceb668749"), True),
    (W_A, H("0522a64d7b931e21760cf955a15fcc733e8a52b42a56ab03afddec8beb668749"), True),
    (W_A[:31], b"\xff" * 32, True), (b"\xff" * 32, W_A[:31], True),
    (H("0522a64d7b931e21760cf955a15fcc733e8a52b42a56ab03afddec8beb668749"), H("0522a64d7b931e21760cf955a15fcc733e8a52b42a56ab03afddec8beb668749"), False),
    (H("0522a64d7b931e213e8a52b42a56ab030522a64d7b931e213e8a52b42a56ab03760cf955a15fcc790522a64d7b931e"), W_A[:31], True),
    (W_A[:31], H("0522a64d7b931e213e8a52b42a56ab030522a64d7b931e213e8a52b42a56ab03760cf955a15fcc790522a64d7b931e"), True)])
def test_inequality(ctx, left, right, ok):
    right_scalars = bpg.be_to_scalars(right)

    def bp(p):
        ls, coms, vars_ = bpg.commit(p, left, [rs(b"ne", i) for i in range(4)])
        g = bpg.Inequality(right_scalars, right_scalars)
        dc, dw = g.setup(p, ls, [rs(b"nd", i) for i in range(2 * len(ls) + 1)])
        g.prove(p, vars_, dw)
        return coms + dc

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        k = len(bpg.be_to_scalars(left))
        bpg.Inequality(right_scalars, None).verify(v, vs[:k], vs[k:])
    run(ctx, b"Inequality", bp, bv, 16, ok)


@pytest.mark.parametrize("left,right,ok", [(3, 5, True), (5, 3, False), (7, 7, False), (0, 1, True), (2**126 - 2, 2**126 - 1, True),
                                           (5, 2**126 + 5, False), (2**125, 2**125 + 2**60, True)])
def test_less_than(ctx, left, right, ok):
    lb, rb = left.to_bytes(17, "big"), right.to_bytes(17, "big")

    def bp(p):
        ls, lc, lv = bpg.commit_single(p, lb, rs(b"lt", 0))
        rs_, rc, rv = bpg.commit_single(p, rb, rs(b"lt", 1))
        g = bpg.LessThan(lv, ls, rv, rs_)
        dc, dw = g.setup(p, [], [rs(b"lt", 2), rs(b"lt", 3)])
        g.prove(p, [], dw)
        return [lc, rc] + dc

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        bpg.LessThan(vs[0], None, vs[1], None).verify(v, [], vs[2:])
    run(ctx, b"LessThan", bp, bv, 512, ok)


LT_A, LT_B = H("0522a64d7b931e21760cf955a15fcc"), H("aa22a64d7b931e21760cf955a15fcc")
LT_MAX, LT_MAXM1 = H("3f" + "ff" * 15), H("3f" + "ff" * 14 + "fe")


@pytest.mark.parametrize("left,right,ok", [(LT_A, LT_B, True), (LT_B, LT_A, False), (LT_MAXM1, LT_MAX, True), (LT_MAX, LT_MAXM1, False),
                                           (b"\x00", b"\x00", False), (LT_MAX, LT_MAX, False)])
def test_less_than_reference_vectors(ctx, left, right, ok):
    """src/less_than/less_than_gadget.rs:96-333 (test_less_than_gadget_1..6): both sides are CONSTANT linear combinations
    (Scalar::into()), no witness commitments, 1024 generators; cases 2, 4, 5, 6 must not verify."""
    ls, rsc = bpg.be_to_scalar(left), bpg.be_to_scalar(right)

    def bp(p):
        g = bpg.LessThan(ls, ls, rsc, rsc)
        dc, dw = g.setup(p, [], [rs(b"ltr", 2), rs(b"ltr", 3)])
        g.prove(p, [], dw)
        return dc

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        bpg.LessThan(ls, None, rsc, None).verify(v, [], vs)
    run(ctx, b"LessThan", bp, bv, 1024, ok)


@pytest.mark.parametrize("member,ok", [(b"\x43", True), (b"\x11", True), (b"\x64", True), (b"\x44", False)])
def test_set_membership(ctx, member, ok):
    inst_set = [bpg.be_to_scalar(b"\x11"), bpg.be_to_scalar(b"\x64")]       # example.gadgets:8 style: instances and witnesses mixed
    wit_set = [b"\x43", b"\x0e\x44"]

    def bp(p):
        ms, mc, mv = bpg.commit_single(p, member, rs(b"sm", 0))
        ws, wc, wv = bpg.commit_all_single(p, wit_set, [rs(b"sm", 1), rs(b"sm", 2)])
        g = bpg.SetMembership(mv, ms, inst_set, inst_set)
        dc, dw = g.setup(p, ws, [rs(b"sd", i) for i in range(4)])
        g.prove(p, wv, dw)
        return [mc] + wc + dc

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        bpg.SetMembership(vs[0], None, inst_set, None).verify(v, vs[1:3], vs[3:])
    run(ctx, b"SetMembership", bp, bv, 8, ok)


SM_V = [None, H("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749"), H("07faf8aaa21077200a11576b1cdb402f52a47f192b36998b4da25807a9be52f5"),
        H("09243333e374e76e4975ab48ae38241ba67805cd60f1523e9b79a48daac9a84d"), H("0258647e47e8005748d4e7d0d76b230cc20f2a0f8745eee2bccced0c2add59d5"),
        H("011c6fc7f15087f4d3e97e672813af066f74f60446bc75aa85eb2d6db8ae791b")]
Z = b"\x00"


@pytest.mark.parametrize("member,wit_set,inst_set,ok", [
    (SM_V[1], [], [SM_V[4], SM_V[3], SM_V[1], SM_V[5], SM_V[2]], True),              # 1: set of instance variables
    (SM_V[1], [SM_V[3], SM_V[5], SM_V[1]], [SM_V[4], SM_V[2]], True),                # 2: mixed set
    (SM_V[1], [SM_V[3], SM_V[5]], [SM_V[4], SM_V[2]], False),                        # 3: mixed set, the value is not a member
    (SM_V[1], [SM_V[3], SM_V[5], Z, SM_V[2]], [SM_V[4], SM_V[2]], False),            # 4: mixed set, a witness is 0 (and the value absent)
    (SM_V[1], [SM_V[3], SM_V[1], SM_V[5]], [SM_V[4], SM_V[2], SM_V[1]], False),      # 5: the value is contained twice (the reference rejects: two indicator bits)
    (Z, [SM_V[3], SM_V[5], Z, SM_V[1]], [SM_V[4], SM_V[2]], True)])                  # 6: zero member
def test_set_membership_reference_vectors(ctx, member, wit_set, inst_set, ok):
    """src/set_membership/set_membership_gadget.rs:174-403 (test_set_membership_gadget_1..6) with the reference's VALUE1..5, 64 generators."""
    inst_sc = [bpg.be_to_scalar(x) for x in inst_set]

    def bp(p):
        ms, mc, mv = bpg.commit_single(p, member, rs(b"smr", 0))
        ws, wc, wv = bpg.commit_all_single(p, wit_set, [rs(b"smr", 1 + i) for i in range(len(wit_set))]) if wit_set else ([], [], [])
        g = bpg.SetMembership(mv, ms, inst_sc, inst_sc)
        dc, dw = g.setup(p, ws, [rs(b"smd", i) for i in range(len(wit_set) + len(inst_set))])
        g.prove(p, wv, dw)
        return [mc] + wc + dc

    def bv(v, coms):
        vs = bpg.verifier_commit(v, coms)
        k = len(wit_set)
        bpg.SetMembership(vs[0], None, inst_sc, None).verify(v, vs[1:1 + k], vs[1 + k:])
    run(ctx, b"SetMembership", bp, bv, 64, ok)


OR_PRE = [H("38535450433043546f313877615a6a423663"),
          H("54686520717569" "4a76077d4a40bd91551b3a03b1ad8adb2b" "666f78206a756d70" "666f78206a756d70" "73206f7665"),
          H("54686520717569636b2062726f776e20666f78206a756d7073206f7665722074")]
OR_IMG = [H("0d2203069ac15f58172bae1b3af98d8982deef9df37482c1a920b8832ee813a4"),
          H("0fcb21fbf23b968dee8f6b3a511e93e8c5c0eb2f71aa0601111f911c9e42cf06"),
          H("01245409f28ae2f076077d4a40bd91551b3a03b1ad8adb2b1da116d29c60a85c")]


def _or_hash_builders(pre, images):
    """three MimcHash256 clauses recorded in a ConstraintBuffer, then or() into the main prover / verifier"""
    widths = [len(bpg.be_to_scalars(x)) for x in pre]
    layout = {}                                       # clause -> number of commitments (witness scalars + derived)

    def bp(p):
        buf = bpg.ConstraintBuffer(p, True)
        coms = []
        for k, (x, img) in enumerate(zip(pre, images)):
            g = bpg.MimcHash256(bpg.be_to_scalar(img))
            sc_, wc, wv = bpg.commit(p, x, [rs(b"or%d" % k, j) for j in range(widths[k])])
            dc, dw = g.setup(p, sc_, [rs(b"or%d" % k, 10), rs(b"or%d" % k, 11)])
            g.prove(buf, wv, dw)
            buf.rewind()
            layout[k] = len(wc) + len(dc)
            coms += wc + dc
        bpg.or_conjunction(p, buf)
        return coms

    def bv(v, coms):
        buf = bpg.ConstraintBuffer(v, False)
        vs = bpg.verifier_commit(v, coms)
        pos = 0
        for k, img in enumerate(images):
            bpg.MimcHash256(bpg.be_to_scalar(img)).verify(buf, vs[pos:pos + widths[k]], vs[pos + widths[k]:pos + layout[k]])
            pos += layout[k]
            buf.rewind()
        bpg.or_conjunction(v, buf)
    return bp, bv


def test_or_conjunction_reference_vectors(ctx):
    """reference src/or/or_conjunction.rs:84-190 (test_or_conjunction_1): three MimcHash256 clauses over the reference's preimages and
    images, 8192 generators; the disjunction verifies.  Each clause contributes 2 explicit constraints (the other 1944+ come from
    multiply), so or() adds 2*2*2 products of two multipliers each."""
    bp, bv = _or_hash_builders(OR_PRE, OR_IMG)
    truth = [bpg.mimc_hash(x) == bpg.be_to_scalar(i) for x, i in zip(OR_PRE, OR_IMG)]
    assert any(truth)
    run(ctx, b"MiMCHash", bp, bv, 8192, True)
    # every image wrong: no clause holds, the proof must not verify
    wrong = [bytes([i[0] ^ 1]) + i[1:] for i in OR_IMG]
    bp, bv = _or_hash_builders(OR_PRE, wrong)
    run(ctx, b"MiMCHash", bp, bv, 8192, False)
    # exactly one true clause, in each position
    for k in range(3):
        imgs = [bpg.scalar_to_be(bpg.mimc_hash(x)) if j == k else wrong[j] for j, x in enumerate(OR_PRE)]
        bp, bv = _or_hash_builders(OR_PRE, imgs)
        run(ctx, b"MiMCHash", bp, bv, 8192, True)


def test_or_conjunction_of_bounds_clauses(ctx):
    """true / false matrix on a gadget with many explicit constraints per clause: BOUND (35) x BOUND = 1,225 products"""
    lo, hi = bytes([10]), bytes([100])

    def build_bounds(values):
        def bp(p):
            buf = bpg.ConstraintBuffer(p, True)
            coms = []
            for k, val in enumerate(values):
                g = bpg.BoundsCheck(lo, hi)
                sc_, wc, wv = bpg.commit(p, bytes([val]), [rs(b"ob%d" % k, 0)])
                dc, dw = g.setup(p, sc_, [rs(b"ob%d" % k, 1), rs(b"ob%d" % k, 2)])
                g.prove(buf, wv, dw)
                buf.rewind()
                coms += wc + dc
            bpg.or_conjunction(p, buf)
            return coms

        def bv(v, coms):
            buf = bpg.ConstraintBuffer(v, False)
            vs = bpg.verifier_commit(v, coms)
            for k in range(len(values)):
                bpg.BoundsCheck(lo, hi).verify(buf, [vs[3 * k]], vs[3 * k + 1:3 * k + 3])
                buf.rewind()
            bpg.or_conjunction(v, buf)
        return bp, bv
    for values, ok in (([50, 5], True), ([5, 50], True), ([50, 60], True), ([5, 200], False)):
        bp, bv = build_bounds(values)
        run(ctx, b"OrBounds", bp, bv, 2048, ok)
