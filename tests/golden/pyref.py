"""Pure-Python (big-int) restatement of the primitives under the Bulletproofs R1CS prover.

TEST INFRASTRUCTURE ONLY.  Used by gen_golden.py to produce the committed fixtures in this
directory, and by the CPU tests as a second, independent check of the C oracle on small inputs.
Nothing here is imported by the product package.

The arithmetic lives in crates that are NOT vendored under /root/reference (curve25519-dalek 1.x,
merlin 1.x, bulletproofs fork; Cargo.toml:8-20), so this follows their published algorithms:
  * Ristretto255: RFC 9496 sections 4.3.1 (decode), 4.3.2 (encode), 4.3.4 (one-way map)
  * STROBE-128 / Merlin v1.0 transcript and TranscriptRng (merlin 1.x strobe.rs / transcript.rs)
  * PedersenGens::default, BulletproofGens::new (dalek bulletproofs generators.rs)
The reference's own call sites: src/bin/prover.rs:52-54,92 ; src/commitments.rs:27 ; src/conversions.rs:18.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
INVSQRT_A_MINUS_D = 54469307008909316920995813868745141605393597292927456921205312896311721017578
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) ** 2 % P


def is_neg(x):
    return (x % P) & 1


def fabs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 4.2 SQRT_RATIO_M1."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = fabs(r)
    return (correct or flipped), r


class Point:
    """Extended twisted-Edwards coordinates (X:Y:Z:T), a = -1."""
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    @staticmethod
    def identity():
        return Point(0, 1, 1, 0)

    def __add__(self, o):
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Point(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Point(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def double(self):
        return self + self

    def __mul__(self, k):
        k %= L
        acc = Point.identity()
        q = self
        while k:
            if k & 1:
                acc = acc + q
            q = q.double()
            k >>= 1
        return acc

    __rmul__ = __mul__

    def __eq__(self, o):
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.X * o.X - self.Y * o.Y) % P == 0

    def compress(self):
        """RFC 9496 4.3.2 Encode."""
        X, Y, Z, T = self.X, self.Y, self.Z, self.T
        u1 = (Z + Y) * (Z - Y) % P
        u2 = X * Y % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1 = invsqrt * u1 % P
        den2 = invsqrt * u2 % P
        z_inv = den1 * den2 % P * T % P
        ix = X * SQRT_M1 % P
        iy = Y * SQRT_M1 % P
        ench = den1 * INVSQRT_A_MINUS_D % P
        rotate = is_neg(T * z_inv)
        if rotate:
            x, y, den_inv = iy, ix, ench
        else:
            x, y, den_inv = X, Y, den2
        if is_neg(x * z_inv):
            y = -y
        s = fabs(den_inv * (Z - y))
        return s.to_bytes(32, "little")


def decompress(b):
    """RFC 9496 4.3.1 Decode. Returns None on failure."""
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    ok, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = fabs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if (not ok) or is_neg(t) or y == 0:
        return None
    return Point(x, y, 1, t)


def elligator(r0):
    """RFC 9496 4.3.4 MAP (dalek: RistrettoPoint::elligator_ristretto_flavor)."""
    r = SQRT_M1 * r0 % P * r0 % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    c = P - 1
    v = (c - r * D) * (r + D) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-fabs(s * r0)) % P
    if not was_square:
        s = s_prime
        c = r
    n = (c * (r - 1) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = n * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return Point(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b64):
    r1 = int.from_bytes(b64[:32], "little") & ((1 << 255) - 1)
    r2 = int.from_bytes(b64[32:], "little") & ((1 << 255) - 1)
    return elligator(r1 % P) + elligator(r2 % P)


BASEPOINT = decompress(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"))


def pedersen_gens():
    """PedersenGens::default(): B = basepoint, B_blinding = hash_from_bytes::<Sha3_512>(compress(B))."""
    B = BASEPOINT
    Bb = from_uniform_bytes(hashlib.sha3_512(B.compress()).digest())
    return B, Bb


def bp_gens(count, party=0):
    """BulletproofGens::new(cap, 1) share 0: SHAKE256("GeneratorsChain" || 'G'|'H' || u32le(party))."""
    out = []
    for tag in (b"G", b"H"):
        stream = hashlib.shake_256(b"GeneratorsChain" + tag + party.to_bytes(4, "little")).digest(64 * count)
        out.append([from_uniform_bytes(stream[64 * i:64 * i + 64]) for i in range(count)])
    return out


# ---------------------------------------------------------------- Keccak-f[1600] / STROBE / Merlin
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
       0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
       0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
       0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M64 if n else v


def keccak_f(state: bytearray):
    A = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        Dd = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ Dd[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = (A[x][y] & _M64).to_bytes(8, "little")


class Strobe128:
    R = 166
    I, A, C, T, M, K = 1, 2, 4, 8, 16, 32

    def __init__(self, label=None):
        if label is None:
            return
        self.st = bytearray(200)
        self.st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        self.st[6:18] = b"STROBEv1.0.2"
        keccak_f(self.st)
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(label, False)

    def clone(self):
        c = Strobe128()
        c.st = bytearray(self.st)
        c.pos, c.pos_begin, c.cur_flags = self.pos, self.pos_begin, self.cur_flags
        return c

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        keccak_f(self.st)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.st[self.pos] = b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert not (flags & self.T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if (flags & (self.C | self.K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(self.M | self.A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(self.A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(self.I | self.A | self.C, more)
        return self._squeeze(n)

    def key(self, data, more):
        self._begin_op(self.A | self.C, more)
        self._overwrite(data)

    def state_bytes(self):
        return bytes(self.st) + bytes([self.pos, self.pos_begin, self.cur_flags])


class Transcript:
    def __init__(self, label):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.strobe.ad(msg, False)

    def append_u64(self, label, v):
        self.append_message(label, v.to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)

    def challenge_scalar(self, label):
        return int.from_bytes(self.challenge_bytes(label, 64), "little") % L

    def build_rng(self, witnesses, seed32):
        """TranscriptRngBuilder: rekey_with_witness_bytes(label, w) for each, then finalize(32 external bytes)."""
        s = self.strobe.clone()
        for label, w in witnesses:
            s.meta_ad(label, False)
            s.meta_ad(len(w).to_bytes(4, "little"), True)
            s.key(w, False)
        s.meta_ad(b"rng", False)
        s.key(seed32, False)
        return TranscriptRng(s)


class TranscriptRng:
    def __init__(self, strobe):
        self.strobe = strobe

    def fill_bytes(self, n):
        self.strobe.meta_ad(n.to_bytes(4, "little"), False)
        return self.strobe.prf(n, False)

    def random_scalar(self):
        return int.from_bytes(self.fill_bytes(64), "little") % L


# ---------------------------------------------------------------- MiMC (reference src/mimc_hash/mimc.rs:7-97)
def scalars_be(data: bytes):
    """conversions.rs:26-30 be_to_scalars: reverse, zero-pad to 32 B multiple, split LE, from_bits (clear bit 255)."""
    b = bytes(reversed(data))
    if len(b) % 32:
        b += bytes(32 - len(b) % 32)
    return [int.from_bytes(b[i:i + 32], "little") & ((1 << 255) - 1) for i in range(0, len(b), 32)]


def mimc_pad(blocks):
    """mimc.rs:77-97 pad(): PKCS#7 on the LE bytes of the last block with trailing zeros stripped."""
    last = blocks[-1].to_bytes(32, "little").rstrip(b"\x00")
    if len(last) < 32:
        k = 32 - len(last)
        padded = int.from_bytes(last + bytes([k]) * k, "little") & ((1 << 255) - 1)
        return blocks[:-1] + [padded]
    return blocks + [int.from_bytes(bytes([32]) * 32, "little") & ((1 << 255) - 1)]


def mimc_encrypt(p, k, consts):
    s = p
    for c in consts:
        t = (s + k + c) % L
        s = t * t % L * t % L
    return (s + k) % L


def mimc_sponge(blocks, consts):
    s = 0
    for b in blocks:
        s = mimc_encrypt((s + b) % L, 0, consts)
    return s


def mimc_hash(data: bytes, consts):
    return mimc_sponge(mimc_pad(scalars_be(data)), consts)
