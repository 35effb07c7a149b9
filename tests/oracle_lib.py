"""ctypes binding of oracle/build/liboracle.so (the CPU checker).  Imported ONLY by tests, smoke() and
bench.py's cpu_baseline leg - never by the product package."""
import ctypes as C
import os
import pathlib
import subprocess

ROOT = pathlib.Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "build" / "liboracle.so"
NATIVE_LIB = ROOT / "oracle" / "build" / "liboracle_native.so"

FLAG_COMPACT_1PHASE = 1
FLAG_NO_1PHASE_DOMSEP = 2
FLAG_EXPANDED_BLINDING = 4      # prover-only opt-in: s_L, s_R expanded from one TranscriptRng draw (include/bpg.h); not upstream's derivation
FLAG_FAST_MSM = 0x100
OK, ERR_GENS_LENGTH, ERR_FORMAT, ERR_VERIFY, ERR_ARG = 0, 1, 2, 3, 4


class Circuit(C.Structure):
    _fields_ = [("n", C.c_uint64), ("q", C.c_uint64), ("m", C.c_uint64), ("nnz", C.c_uint64), ("ncoef", C.c_uint64),
                ("aL", C.c_void_p), ("aR", C.c_void_p), ("aO", C.c_void_p),
                ("row_ptr", C.c_void_p), ("term_var", C.c_void_p), ("term_coef", C.c_void_p), ("coef", C.c_void_p)]


def build():
    if not LIB.exists() or any(p.stat().st_mtime > LIB.stat().st_mtime for p in (ROOT / "oracle" / "src").iterdir()):
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)


_lib = None
_lib_path = None


def use_native():
    """bench.py's cpu_baseline leg: build the oracle with -march=native ON THIS HOST and load that build (before any other use of the
    oracle in the process). Returns the path loaded; falls back to the portable build when the native one cannot be made."""
    global _lib_path
    if _lib is not None:
        return str(_lib_path)
    try:
        NATIVE_LIB.unlink(missing_ok=True)       # never trust a native build that travelled from another machine
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "native"], stdout=subprocess.DEVNULL)
        _lib_path = NATIVE_LIB
    except Exception:
        _lib_path = None
    lib()
    return str(_lib_path)


def lib():
    global _lib, _lib_path
    if _lib is None:
        if _lib_path is None:
            if not LIB.exists():
                build()
            _lib_path = LIB
        _lib = C.CDLL(str(_lib_path))
        _lib.orc_gens_new.restype = C.c_void_p
        _lib.orc_gens_new.argtypes = [C.c_uint64]
        _lib.orc_gens_from_compressed.restype = C.c_void_p
        _lib.orc_gens_from_compressed.argtypes = [C.c_uint64, C.c_char_p, C.c_char_p]
        _lib.orc_gens_free.argtypes = [C.c_void_p]
        _lib.orc_gens_export.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p]
        _lib.orc_r1cs_prove.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Circuit), C.c_char_p, C.c_char_p, C.c_uint32,
                                        C.c_char_p, C.POINTER(C.c_uint64)]
        _lib.orc_r1cs_verify.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Circuit), C.c_char_p, C.c_char_p, C.c_uint64,
                                         C.c_char_p, C.c_uint32]
        _lib.orc_r1cs_satisfied.argtypes = [C.POINTER(Circuit), C.c_char_p]
        _lib.orc_msm.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int]
        _lib.orc_rng_scalars.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p]
        _lib.orc_transcript_init.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_transcript_append.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_transcript_append_u64.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_transcript_challenge.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_transcript_challenge_scalar.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        _lib.orc_mimc_hash.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_mimc_sponge.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_sha3_512.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        _lib.orc_shake256.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    return _lib


def _buf(n):
    return C.create_string_buffer(n)


class Transcript:
    def __init__(self, label=None, state=None):
        self.st = _buf(203)
        if state is not None:
            self.st.raw = bytes(state)
        else:
            lib().orc_transcript_init(self.st, label, len(label))

    def append(self, label, msg):
        lib().orc_transcript_append(self.st, label, msg, len(msg))

    def append_u64(self, label, v):
        lib().orc_transcript_append_u64(self.st, label, v)

    def challenge(self, label, n):
        o = _buf(n)
        lib().orc_transcript_challenge(self.st, label, o, n)
        return o.raw

    def challenge_scalar(self, label):
        o = _buf(32)
        lib().orc_transcript_challenge_scalar(self.st, label, o)
        return o.raw

    @property
    def state(self):
        return self.st.raw


def _bin(name, a, b):
    o = _buf(32)
    getattr(lib(), name)(o, a, b)
    return o.raw


def sc_mul(a, b): return _bin("orc_sc_mul", a, b)
def sc_add(a, b): return _bin("orc_sc_add", a, b)
def sc_sub(a, b): return _bin("orc_sc_sub", a, b)


def sc_invert(a):
    o = _buf(32); lib().orc_sc_invert(o, a); return o.raw


def sc_wide(b64):
    o = _buf(32); lib().orc_sc_wide(o, b64); return o.raw


def sc_reduce(b32):
    o = _buf(32); lib().orc_sc_reduce(o, b32); return o.raw


def from_uniform(b64):
    o = _buf(32); lib().orc_from_uniform(o, b64); return o.raw


def point_mul(k, p):
    o = _buf(32); rc = lib().orc_point_mul(o, k, p); return o.raw if rc == 0 else None


def point_add(p, q):
    o = _buf(32); rc = lib().orc_point_add(o, p, q); return o.raw if rc == 0 else None


def pedersen_bases():
    a, b = _buf(32), _buf(32); lib().orc_pedersen_bases(a, b); return a.raw, b.raw


def pedersen_commit(v, r):
    o = _buf(32); lib().orc_pedersen_commit(o, v, r); return o.raw


def msm(scalars, points, algo=1):
    o = _buf(32)
    rc = lib().orc_msm(o, len(scalars) // 32, scalars, points, algo)
    assert rc == 0
    return o.raw


def mimc_hash(pre):
    o = _buf(32); lib().orc_mimc_hash(o, pre, len(pre)); return o.raw


def mimc_sponge(blocks):
    o = _buf(32); lib().orc_mimc_sponge(o, blocks, len(blocks) // 32); return o.raw


def rng_scalars(tstate, v_blinding, seed, count):
    o = _buf(32 * count)
    lib().orc_rng_scalars(tstate, len(v_blinding) // 32, v_blinding, seed, count, o)
    return [o.raw[32 * i:32 * i + 32] for i in range(count)]


class Gens:
    def __init__(self, capacity=None, compressed=None):
        if compressed is not None:
            G, H = compressed
            self.cap = len(G) // 32
            self.h = lib().orc_gens_from_compressed(self.cap, G, H)
            assert self.h, "generator decompression failed"
        else:
            self.cap = capacity
            self.h = lib().orc_gens_new(capacity)

    def export(self, first, count):
        g, h = _buf(32 * count), _buf(32 * count)
        lib().orc_gens_export(self.h, first, count, g, h)
        return g.raw, h.raw

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # module globals are already torn down at interpreter exit
            lib().orc_gens_free(self.h)
            self.h = None


class FlatCircuit:
    """Flattened R1CS instance in the layout both the oracle and include/bpg.h consume.
    aL/aR/aO: bytes (n*32); row_ptr: array('Q'); term_var/term_coef: array('I'); coef: bytes (ncoef*32)."""

    def __init__(self, n, m, aL, aR, aO, row_ptr, term_var, term_coef, coef):
        import numpy as np
        self.n, self.m = n, m
        self.aL, self.aR, self.aO = aL, aR, aO
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.term_var = np.ascontiguousarray(term_var, dtype=np.uint32)
        self.term_coef = np.ascontiguousarray(term_coef, dtype=np.uint32)
        self.coef = coef
        self.q = len(self.row_ptr) - 1
        self.nnz = len(self.term_var)
        self.ncoef = len(coef) // 32

    def cstruct(self):
        c = Circuit()
        c.n, c.q, c.m, c.nnz, c.ncoef = self.n, self.q, self.m, self.nnz, self.ncoef
        self._keep = [C.create_string_buffer(x, len(x)) if x is not None else None for x in (self.aL, self.aR, self.aO, self.coef)]
        c.aL, c.aR, c.aO = [C.cast(k, C.c_void_p).value if k is not None else None for k in self._keep[:3]]
        c.coef = C.cast(self._keep[3], C.c_void_p).value
        c.row_ptr = self.row_ptr.ctypes.data
        c.term_var = self.term_var.ctypes.data
        c.term_coef = self.term_coef.ctypes.data
        return c


def proof_size(n, flags=0):
    N = 1
    while N < n:
        N *= 2
    lg = N.bit_length() - 1
    return (1 + 11 * 32 if flags & FLAG_COMPACT_1PHASE else 14 * 32) + (2 * lg + 2) * 32


def prove(gens, tstate, circ, v_blinding, seed, flags=0):
    """Returns (rc, proof_bytes, tstate_after)."""
    ts = _buf(203); ts.raw = bytes(tstate)
    cap = proof_size(circ.n, flags)
    out = _buf(cap)
    ln = C.c_uint64(cap)
    cs = circ.cstruct()
    rc = lib().orc_r1cs_prove(gens.h, ts, C.byref(cs), v_blinding, seed, flags, out, C.byref(ln))
    return rc, out.raw[:ln.value], ts.raw


def verify(gens, tstate, circ, V, proof, seed=bytes(32), flags=0):
    ts = _buf(203); ts.raw = bytes(tstate)
    cs = circ.cstruct()
    return lib().orc_r1cs_verify(gens.h, ts, C.byref(cs), V, proof, len(proof), seed, flags)


def satisfied(circ, v):
    cs = circ.cstruct()
    return bool(lib().orc_r1cs_satisfied(C.byref(cs), v))
