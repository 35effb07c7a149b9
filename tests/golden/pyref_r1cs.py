"""Pure-Python restatement of the reference's R1CS ASSEMBLY: the recording constraint system (what bulletproofs' Prover keeps when
gadgets call multiply / allocate_multiplier / constrain) and the seven gadgets, read from the reference text.

TEST INFRASTRUCTURE ONLY (tests/, tests/golden/gen_assembly_fixtures.py).  Independent of the product: big-int arithmetic, no ctypes, no
shared library.  It exists so that constraint ORDER and the terms of every linear combination - which decide the z^j weights and hence the
proof bytes - are checked against a second implementation, not only by counts and by "it verifies".

Follows, line by line:
  utils::range_proof                       /root/reference/src/utils.rs:5-35
  BoundsCheck                              src/bounds_check/bounds_check_gadget.rs:13-63
  MimcHash256 (pad, sponge, encryption)    src/mimc_hash/mimc_hash_gadget.rs:12-150
  MerkleTree256::parse                     src/merkle_tree/merkle_tree_gadget.rs:39-113
  Equality / Inequality / LessThan / SetMembership   src/equality/equality_gadget.rs:10-40, src/inequality/inequality_gadget.rs:11-116,
                                           src/less_than/less_than_gadget.rs:15-84, src/set_membership/set_membership_gadget.rs:12-131
  Gadget::setup / prove                    src/gadget.rs:18-47
  commit / commit_single                   src/commitments.rs:22-43
  conversions                              src/conversions.rs:6-73
The constraint-system semantics (multiply = allocate + two constraints `left - l = 0`, `right - r = 0`, in that order) are those of the
un-vendored bulletproofs crate's r1cs::Prover, visible from the reference at src/cs_buffer.rs:89-113.
"""
import hashlib

import pyref as R

L = R.L
KIND_L, KIND_R, KIND_O, KIND_V, KIND_ONE = 0, 1, 2, 3, 4


def var(kind, idx=0):
    return (kind << 29) | idx


ONE = var(KIND_ONE)


class LC:
    """bulletproofs::r1cs::LinearCombination: a list of (Variable, Scalar) terms; arithmetic appends / scales terms, nothing is merged."""
    __slots__ = ("terms",)

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def of(x):
        if isinstance(x, LC):
            return LC(x.terms)
        if isinstance(x, (bytes, bytearray)):               # a Scalar constant: (One, s)
            return LC([(ONE, int.from_bytes(x, "little") % L)])
        if isinstance(x, bool):
            raise TypeError(x)
        if isinstance(x, int):                               # a packed Variable
            return LC([(int(x), 1)])
        raise TypeError(type(x))

    @staticmethod
    def const(s):
        return LC([(ONE, s % L)])

    def __add__(self, o):
        return LC(self.terms + LC.of(o).terms)

    def __sub__(self, o):
        return LC(self.terms + [(v, (-c) % L) for v, c in LC.of(o).terms])

    def __neg__(self):
        return LC([(v, (-c) % L) for v, c in self.terms])

    def scale(self, s):
        return LC([(v, c * s % L) for v, c in self.terms])


def sc_int(b):
    return int.from_bytes(b, "little")


def sc_bytes(x):
    return (x % L).to_bytes(32, "little")


def be_to_scalars(data: bytes):
    """conversions::be_to_scalars (conversions.rs:26-30): list of raw from_bits integers (bit 255 cleared, NOT reduced)."""
    return R.scalars_be(bytes(data))


def be_to_scalar(data: bytes):
    assert len(data) <= 32, "the given vector is longer than 32 bytes"
    return R.scalars_be(bytes(data) if data else b"\x00")[0]


class Transcript:
    """merlin::Transcript::new(label)"""
    def __init__(self, label: bytes):
        self.label = bytes(label)
        self.t = R.Transcript(self.label)

    @property
    def state(self):
        return self.t.strobe.state_bytes()


class RecordingProver:
    """bulletproofs::r1cs::Prover as far as assembly goes: commitments (with their "V" appends), witness vectors, constraint list."""

    def __init__(self, ctx, transcript: Transcript):
        self.transcript = transcript
        transcript.t.append_message(b"dom-sep", b"r1cs v1")          # Prover::new -> transcript.r1cs_domain_sep()
        self.aL, self.aR, self.aO, self.v, self.vb = [], [], [], [], []
        self.constraints = []
        self.B, self.Bb = R.pedersen_gens()

    # -- Prover::commit(v, v_blinding): V = v*B + r*B_blinding, transcript.append_point("V", V)
    def commit(self, v, v_blinding):
        vi = sc_int(v) if isinstance(v, (bytes, bytearray)) else int(v)
        ri = sc_int(v_blinding) % L
        com = (self.B * (vi % L) + self.Bb * ri).compress()
        self.transcript.t.append_message(b"V", com)
        self.v.append(vi); self.vb.append(ri)
        return com, var(KIND_V, len(self.v) - 1)

    def commit_many(self, vs, blindings):
        out = [self.commit(v, b) for v, b in zip(vs, blindings)]
        return [c for c, _ in out], [x for _, x in out]

    # -- ConstraintSystem
    def eval(self, lc):
        acc = 0
        for v, c in LC.of(lc).terms:
            kind, idx = v >> 29, v & 0x1fffffff
            x = (self.aL[idx] if kind == KIND_L else self.aR[idx] if kind == KIND_R else self.aO[idx] if kind == KIND_O
                 else self.v[idx] if kind == KIND_V else 1)
            acc = (acc + c * x) % L
        return acc

    def multiply(self, left, right):
        left, right = LC.of(left), LC.of(right)
        l, r = self.eval(left), self.eval(right)
        i = len(self.aL)
        self.aL.append(l); self.aR.append(r); self.aO.append(l * r % L)
        lv, rv, ov = var(KIND_L, i), var(KIND_R, i), var(KIND_O, i)
        self.constrain(left - lv)
        self.constrain(right - rv)
        return lv, rv, ov

    def allocate_multiplier(self, assignment):
        if assignment is None:
            raise ValueError("missing assignment")
        l, r = (sc_int(x) if isinstance(x, (bytes, bytearray)) else int(x) for x in assignment)
        i = len(self.aL)
        self.aL.append(l % L); self.aR.append(r % L); self.aO.append(l * r % L)
        return var(KIND_L, i), var(KIND_R, i), var(KIND_O, i)

    def constrain(self, lc):
        self.constraints.append(LC.of(lc))

    def num_constraints(self):
        return len(self.constraints)

    def get_num_multiplications(self):
        return len(self.aL)

    def prove(self, *a, **k):                                 # assembly only
        return b""

    def satisfied(self):
        return all(self.eval(c) == 0 for c in self.constraints)


def canonical_rows(rows):
    """What decides the flattened weights: per constraint, in order, the coefficient of every variable (terms of one variable added up
    mod l, zero coefficients dropped, variables sorted).  rows: iterable of iterables of (packed variable, coefficient int)."""
    out = []
    for terms in rows:
        acc = {}
        for v, c in terms:
            acc[v] = (acc.get(v, 0) + c) % L
        out.append(sorted((v, c) for v, c in acc.items() if c))
    return out


def digest_rows(rows):
    h = hashlib.sha256()
    for row in canonical_rows(rows):
        h.update(len(row).to_bytes(4, "little"))
        for v, c in row:
            h.update(v.to_bytes(4, "little")); h.update(c.to_bytes(32, "little"))
    return h.hexdigest()


def digest_scalars(*vectors):
    h = hashlib.sha256()
    for vec in vectors:
        h.update(len(vec).to_bytes(8, "little"))
        for x in vec:
            h.update((x % L).to_bytes(32, "little"))
    return h.hexdigest()


def summary(p: RecordingProver):
    return {"n": len(p.aL), "q": len(p.constraints), "m": len(p.v), "constraints_sha256": digest_rows(c.terms for c in p.constraints),
            "witness_sha256": digest_scalars(p.aL, p.aR, p.aO), "committed_sha256": digest_scalars(p.v, p.vb)}


# ------------------------------------------------------------------------------------------------ utils.rs
def range_proof(cs, x, n_bits, x_assignment=None):
    """utils::range_proof (src/utils.rs:5-35)"""
    x = LC.of(x)
    exp_2 = 1
    xb = None
    if x_assignment is not None:
        xb = x_assignment if isinstance(x_assignment, (bytes, bytearray)) else int(x_assignment).to_bytes(32, "little")
    for i in range(n_bits):
        if xb is None:
            a, b, o = cs.allocate_multiplier(None)
        else:
            bit = (xb[i // 8] >> (i % 8)) & 1
            a, b, o = cs.allocate_multiplier((1 - bit, bit))
        cs.constrain(LC.of(o))                                # a * b = 0
        cs.constrain(LC.of(a) + (LC.of(b) - LC.const(1)))     # a + (b - 1)
        x = x - LC.of(b).scale(exp_2)
        exp_2 = exp_2 * 2 % L
    cs.constrain(x)


# ------------------------------------------------------------------------------------------------ gadget.rs
class Gadget:
    def preprocess(self, witnesses):
        return []

    def setup(self, prover, witnesses, blindings):
        """Gadget::setup (src/gadget.rs:18-38): one commitment per derived scalar -> (commitments, [(scalar bytes, Variable)])"""
        ws = [sc_int(w) if isinstance(w, (bytes, bytearray)) else int(w) for w in witnesses]
        derived = self.preprocess(ws)
        coms, out = [], []
        for s, blind in zip(derived, blindings):
            com, v = prover.commit(s, blind)
            coms.append(com); out.append((s.to_bytes(32, "little"), v))
        assert len(out) == len(derived), "too few blinding factors"
        return coms, out

    def prove(self, cs, commitment_vars, derived_witnesses):
        self.assemble(cs, list(commitment_vars), [(sc_int(s) if s is not None else None, v) for s, v in derived_witnesses])


class BoundsCheck(Gadget):
    def __init__(self, min_be, max_be):                       # bounds_check_gadget.rs:54-63
        self.n = (len(max_be) * 8) & 0xff
        self.min, self.max = be_to_scalar(min_be), be_to_scalar(max_be)

    def preprocess(self, witnesses):                          # :14-21
        v = witnesses[0]
        return [(v - self.min) % L, (self.max - v) % L]

    def assemble(self, cs, _, derived):                       # :23-47
        (a_assignment, a), (b_assignment, b) = derived[0], derived[1]
        a_lc, b_lc = LC.of(a), LC.of(b)
        cs.constrain((a_lc + b_lc) - LC.const(self.max - self.min))
        range_proof(cs, a_lc, self.n, a_assignment)
        range_proof(cs, b_lc, self.n, b_assignment)


_RC = None


def round_constants():
    global _RC
    if _RC is None:
        import pathlib
        text = (pathlib.Path(__file__).resolve().parent / "mimc_rc769.hex").read_text().split()
        _RC = [int.from_bytes(bytes.fromhex(h), "little") & ((1 << 255) - 1) for h in text]      # Scalar::from_bits(*constant)
        assert len(_RC) == 486
    return _RC


def mimc_hash(data: bytes):
    return R.mimc_hash(bytes(data), [c % L for c in round_constants()])


class MimcHash256(Gadget):
    ROUNDS, BLOCK = 486, 32

    def __init__(self, image=None):
        self.image = LC.const(0) if image is None else LC.of(image)

    def preprocess(self, witnesses):                          # mimc_hash_gadget.rs:15-37
        last = witnesses[-1]
        le = last.to_bytes(32, "little").rstrip(b"\x00")     # remove_zero_padding!
        if len(le) < self.BLOCK:
            k = self.BLOCK - len(le)
            padded = int.from_bytes(le + bytes([k]) * k, "little") & ((1 << 255) - 1)
            return [padded, (padded - last) % L]
        return [int.from_bytes(bytes([32]) * 32, "little") & ((1 << 255) - 1)]

    def assemble(self, cs, witnesses, derived):               # :39-50
        commitments = self.pad(cs, witnesses, derived)
        hash_lc = self.mimc_sponge(cs, [LC.of(c) for c in commitments])
        cs.constrain(hash_lc - self.image)

    def pad(self, cs, witnesses, derived):                    # :82-107
        commitments = list(witnesses)
        _, padded_block = derived[0]
        if len(derived) == 2:
            _, padding = derived[1]
            last_block = LC.of(commitments.pop())
            cs.constrain((last_block + LC.of(padding)) - LC.of(padded_block))
        commitments.append(padded_block)
        return commitments

    def mimc_sponge(self, cs, preimage):                      # :109-124
        key_zero = LC.const(0)
        state = LC.const(0)
        for variable in preimage:
            state = state + variable
            state = self.mimc_encryption(cs, state, key_zero)
        return state

    def mimc_encryption(self, cs, p, k):                      # :126-150
        rc = round_constants()
        p_v, k_v = p, k
        for i in range(self.ROUNDS):
            t = (p_v + k_v) + LC.const(rc[i])
            x_k_ci, _, sqr = cs.multiply(t, t)
            _, _, cube = cs.multiply(LC.of(sqr), LC.of(x_k_ci))
            p_v = LC.of(cube)
        return p_v + k_v


def parse_pattern(text):
    """"((W I) W)" -> nested tuples ('H', l, r) | 'W' | 'I' (the Pattern enum, merkle_tree_gadget.rs:14-19)"""
    toks = text.replace("(", " ( ").replace(")", " ) ").split()
    pos = 0

    def node():
        nonlocal pos
        t = toks[pos]; pos += 1
        if t == "(":
            l = node(); r = node()
            assert toks[pos] == ")"; pos += 1
            return ("H", l, r)
        assert t in ("W", "I"), t
        return t
    out = node()
    assert pos == len(toks)
    return out


class MerkleTree256(Gadget):
    def __init__(self, root, instance_vars, witness_vars, pattern):
        self.root = LC.of(root)
        self.instance_vars = [LC.of(x) for x in instance_vars]
        self.witness_vars = [LC.of(x) for x in witness_vars]
        self.pattern = parse_pattern(pattern) if isinstance(pattern, str) else pattern
        self.gadget = MimcHash256()

    def assemble(self, cs, _w, _d):                           # merkle_tree_gadget.rs:44-56
        w, i = list(self.witness_vars), list(self.instance_vars)
        h = self.parse(cs, w, i, self.pattern)
        cs.constrain(h - self.root)

    def parse(self, cs, w_vars, i_vars, pattern):             # :75-107: children left to right, a leaf takes the next W / I value
        def nxt(values):
            assert values, "too few variables provided to satisfy the given pattern"
            return values.pop(0)
        if pattern == "W":
            preimage = [nxt(w_vars)]
        elif pattern == "I":
            preimage = [nxt(i_vars)]
        else:
            preimage = []
            for child in pattern[1:]:
                if child == "W":
                    preimage.append(nxt(w_vars))
                elif child == "I":
                    preimage.append(nxt(i_vars))
                else:
                    preimage.append(self.parse(cs, w_vars, i_vars, child))
        return self.gadget.mimc_sponge(cs, preimage)


class Equality(Gadget):
    def __init__(self, right_hand):
        self.right_hand = [LC.of(x) for x in right_hand]

    def assemble(self, cs, left_hand, _):                     # equality_gadget.rs:15-31
        if len(self.right_hand) != len(left_hand):
            return cs.constrain(LC.const(1))
        for r, l in zip(self.right_hand, left_hand):
            cs.constrain(r - LC.of(l))


class Inequality(Gadget):
    def __init__(self, right_hand, right_hand_assignment=None):
        self.right_hand = [LC.of(x) for x in right_hand]
        self.right_assignment = None if right_hand_assignment is None else [sc_int(x) if isinstance(x, (bytes, bytearray)) else int(x) for x in right_hand_assignment]

    @staticmethod
    def compare(left, right):                                 # inequality_gadget.rs:103-113: byte-wise from the top, on as_bytes()
        return left.to_bytes(32, "little")[::-1] >= right.to_bytes(32, "little")[::-1]

    def preprocess(self, left_hand):                          # :12-43
        assert self.right_assignment is not None, "missing right hand assignment"
        out, total = [], 0
        for i, left in enumerate(left_hand):
            right = self.right_assignment[i] if i < len(self.right_assignment) else 0
            delta = (left - right) % L if self.compare(left, right) else (right - left) % L
            out.append(delta)
            if delta == 0:
                out.append(0)
            else:
                inv = pow(delta, L - 2, L)
                out.append(inv)
                total = (total + delta * inv) % L
        out.append(pow(total, L - 2, L))                      # Scalar::invert of zero is zero
        return out

    def assemble(self, cs, left_hand, derived):               # :45-92
        if len(self.right_hand) != len(left_hand):
            return cs.constrain(LC.const(0))
        total = LC.const(0)
        for i in range(len(left_hand)):
            right_lc, left_lc = self.right_hand[i], LC.of(left_hand[i])
            delta, delta_inv = derived[2 * i][1], derived[2 * i + 1][1]
            left = (left_lc - right_lc) - LC.of(delta)
            right = (right_lc - left_lc) - LC.of(delta)
            _, _, should_be_zero = cs.multiply(left, right)
            cs.constrain(LC.of(should_be_zero))
            _, _, zero_or_one = cs.multiply(LC.of(delta), LC.of(delta_inv))
            total = total + LC.of(zero_or_one)
        sum_inv = LC.of(derived[-1][1])
        _, _, should_be_one = cs.multiply(total, sum_inv)
        cs.constrain(LC.const(1) - LC.of(should_be_one))


class LessThan(Gadget):
    def __init__(self, left_hand, left_assignment, right_hand, right_assignment):
        conv = lambda x: None if x is None else (sc_int(x) if isinstance(x, (bytes, bytearray)) else int(x))
        self.left, self.right = LC.of(left_hand), LC.of(right_hand)
        self.left_assignment, self.right_assignment = conv(left_assignment), conv(right_assignment)

    def preprocess(self, _):                                  # less_than_gadget.rs:16-35
        delta = (self.right_assignment - self.left_assignment) % L
        return [delta, 0 if delta == 0 else pow(delta, L - 2, L)]

    def assemble(self, cs, _, derived):                       # :37-67
        (delta_assignment, delta), (_, delta_inv) = derived[0], derived[1]
        delta_lc = LC.of(delta)
        n = 126
        range_proof(cs, self.left, n, self.left_assignment)
        range_proof(cs, self.right, n, self.right_assignment)
        range_proof(cs, delta_lc, n, delta_assignment)
        _, _, should_be_one = cs.multiply(delta_lc, LC.of(delta_inv))
        cs.constrain(LC.const(1) - LC.of(should_be_one))
        cs.constrain((self.right - self.left) - delta_lc)


class SetMembership(Gadget):
    def __init__(self, value, value_assignment, instance_vars, instance_assignments):
        conv = lambda x: sc_int(x) if isinstance(x, (bytes, bytearray)) else int(x)
        self.value = LC.of(value)
        self.value_assignment = None if value_assignment is None else conv(value_assignment)
        self.instance_vars = [LC.of(x) for x in instance_vars]
        self.instance_assignments = None if instance_assignments is None else [conv(x) for x in instance_assignments]

    def preprocess(self, witnesses):                          # set_membership_gadget.rs:13-35: one-hot vector over witnesses ++ instances
        return [1 if e == self.value_assignment else 0 for e in list(witnesses) + list(self.instance_assignments)]

    def assemble(self, cs, witnesses, derived):               # :37-62
        one_hot = []
        for _, bit in derived:
            bit_lc = LC.of(bit)
            _, _, should_be_zero = cs.multiply(LC.const(1) - bit_lc, bit_lc)      # is_bit :103-110
            cs.constrain(LC.of(should_be_zero))
            one_hot.append(bit_lc)
        total = LC.const(0)                                   # one_hot_vector :84-99
        for bit in one_hot:
            total = total + bit
        cs.constrain(LC.const(1) - total)
        elems = [LC.of(w) for w in witnesses] + list(self.instance_vars)
        if len(one_hot) != len(elems):                        # hadamard_product :112-131
            return cs.constrain(LC.const(1))
        actual = LC.const(0)
        for a, b in zip(one_hot, elems):
            _, _, product = cs.multiply(a, b)
            actual = actual + LC.of(product)
        cs.constrain(self.value - actual)


# ------------------------------------------------------------------------------------------------ commitments.rs
def commit_single(prover, witness: bytes, blinding: bytes):
    assert len(witness) <= 32, "the provided witness is longer than 32 bytes"
    s = be_to_scalar(witness)
    com, v = prover.commit(s, blinding)
    return s.to_bytes(32, "little"), com, v


def commit(prover, witness: bytes, blindings):
    scalars = be_to_scalars(witness)
    coms, vars_ = prover.commit_many(scalars, list(blindings)[:len(scalars)])
    return [s.to_bytes(32, "little") for s in scalars], coms, vars_


def commit_all_single(prover, witnesses, blindings):
    scalars = [be_to_scalar(w) for w in witnesses]
    coms, vars_ = prover.commit_many(scalars, list(blindings)[:len(scalars)])
    return [s.to_bytes(32, "little") for s in scalars], coms, vars_
