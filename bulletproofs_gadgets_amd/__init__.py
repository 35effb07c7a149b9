"""bulletproofs_gadgets_amd - MI355X-native Bulletproofs R1CS prove path behind the reference's Gadget/Prover surface.

Thin ctypes binding of libbpg_hip.so (include/bpg.h).  Names mirror the reference so that its tests translate
line by line:
    Transcript, PedersenGens, BulletproofGens, Prover, Verifier          (reference src/bin/prover.rs:47-100)
    commit, commit_single, commit_all_single, verifier_commit             (src/commitments.rs:8-47)
    BoundsCheck, MimcHash256, MerkleTree256, Pattern strings, range_proof (src/*/..._gadget.rs, src/utils.rs)
    be_to_scalar(s), scalar_to_be, mimc_hash                              (src/conversions.rs, src/mimc_hash/mimc.rs)
All arithmetic happens in the shared library; there is no Python or CPU fallback - importing works without a GPU,
creating a Context does not.
"""
import ctypes as C
import os
import pathlib

from . import build as _build

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = _PKG / "libbpg_hip.so"

FLAG_COMPACT_1PHASE = 1
FLAG_NO_1PHASE_DOMSEP = 2
FLAG_EXPANDED_BLINDING = 4      # prover-only opt-in: s_L, s_R expanded from one TranscriptRng draw (include/bpg.h); not upstream's derivation
VAR_MULTIPLIER_LEFT, VAR_MULTIPLIER_RIGHT, VAR_MULTIPLIER_OUTPUT, VAR_COMMITTED, VAR_ONE = 0, 1, 2, 3, 4
L = 2**252 + 27742317777372353535851937790883648493

STATUS_NAMES = {0: "OK", 1: "INVALID_GENERATORS_LENGTH", 2: "FORMAT_ERROR", 3: "VERIFICATION_ERROR", 4: "INVALID_ARGUMENT",
                5: "MISSING_ASSIGNMENT", 6: "GADGET_ERROR", 7: "DEVICE_ERROR", 8: "INTERNAL"}


class BpgError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class R1CSInstance(C.Structure):
    _fields_ = [("n", C.c_uint64), ("q", C.c_uint64), ("m", C.c_uint64), ("nnz", C.c_uint64), ("ncoef", C.c_uint64),
                ("aL", C.c_void_p), ("aR", C.c_void_p), ("aO", C.c_void_p), ("row_ptr", C.c_void_p),
                ("term_var", C.c_void_p), ("term_coef", C.c_void_p), ("coef", C.c_void_p)]


class Timings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rng_host", "msm_aiao", "msm_s", "poly", "ipa", "total", "ipa_msm", "ipa_fold", "ipa_sync")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Config(C.Structure):
    """bpg_config (include/bpg.h): zero / None fields fall back to the BPG_* environment variable, then to the profile's default."""
    _fields_ = [("struct_size", C.c_uint32), ("profile", C.c_uint32), ("table_budget_gb", C.c_double), ("chain_workers", C.c_uint32),
                ("chain_lanes", C.c_uint32), ("blocking_sync", C.c_int32), ("gens_cache_dir", C.c_char_p)]


PROFILE_DEFAULT, PROFILE_ONESHOT, PROFILE_SERVING = 0, 1, 2
_PROFILES = {None: 0, "default": 0, "oneshot": 1, "one-shot": 1, "serving": 2, 0: 0, 1: 1, 2: 2}


def make_config(profile=None, table_budget_gb=None, chain_workers=None, chain_lanes=None, blocking_sync=None, gens_cache_dir=None):
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.profile = _PROFILES[profile]
    cfg.table_budget_gb = float(table_budget_gb or 0)
    cfg.chain_workers, cfg.chain_lanes = int(chain_workers or 0), int(chain_lanes or 0)
    cfg.blocking_sync = -1 if blocking_sync is None else (1 if blocking_sync else 0)      # -1 unset (environment, else spin), 1 blocking, 0 spin
    cfg.gens_cache_dir = os.fsencode(gens_cache_dir) if gens_cache_dir else None
    return cfg


class _Term(C.Structure):
    _pack_ = 1
    _fields_ = [("var", C.c_uint32), ("coeff", C.c_uint8 * 32)]


class _LC(C.Structure):
    _fields_ = [("terms", C.POINTER(_Term)), ("n", C.c_uint64)]


_lib = None


def lib():
    """Load (building first if the sources are newer) the shared library."""
    global _lib
    if _lib is None:
        # BPG_LIB_PATH: another build of the SAME sources - the sanitizer builds of tests/hostcheck (host side under ASan/UBSan or TSan) - for the
        # device-less tests; never a different implementation
        override = os.environ.get("BPG_LIB_PATH")
        if override:
            _lib = C.CDLL(override)
        else:
            if not LIB_PATH.exists() or (os.environ.get("BPG_REBUILD") and _build.needs_build()):
                _build.build()
            _lib = C.CDLL(str(LIB_PATH))
        _lib.bpg_strerror.restype = C.c_char_p
        _lib.bpg_last_error.restype = C.c_char_p
        for name in ("bpg_proof_size", "bpg_prover_num_constraints", "bpg_prover_num_multiplications", "bpg_prover_num_committed",
                     "bpg_verifier_num_vars"):
            getattr(_lib, name).restype = C.c_uint64
        _lib.bpg_proof_size.argtypes = [C.c_uint64, C.c_uint32]
    return _lib


def _chk(status):
    if status != 0:
        raise BpgError(status, lib().bpg_last_error().decode())


def _buf(n):
    return C.create_string_buffer(max(n, 1))


def _seed32(seed):
    """The 32 bytes that stand in for upstream's thread_rng() draw: fresh OS randomness unless the caller pins them (tests, benchmarks).
    A constant seed makes every blinding factor a function of the public transcript and the commitment blindings alone."""
    if seed is None:
        return os.urandom(32)
    seed = bytes(seed)
    if len(seed) != 32:
        raise ValueError("rng seed must be exactly 32 bytes")
    return seed


def _exact(name, data, nbytes):
    data = bytes(data)
    if len(data) != nbytes:
        raise ValueError("%s must be exactly %d bytes, got %d" % (name, nbytes, len(data)))
    return data


# ------------------------------------------------------------------------------------------------ scalars / conversions
def scalar_from_int(x):
    return (x % L).to_bytes(32, "little")


def be_to_scalars(data: bytes):
    """conversions::be_to_scalars (src/conversions.rs:26-30): list of 32-byte little-endian Scalar encodings."""
    n = C.c_uint64(len(data) // 32 + 2)
    out = _buf(32 * n.value)
    _chk(lib().bpg_be_to_scalars(bytes(data), C.c_uint64(len(data)), out, C.byref(n)))
    return [out.raw[32 * i:32 * i + 32] for i in range(n.value)]


def be_to_scalar(data: bytes):
    """conversions::be_to_scalar (src/conversions.rs:49-53)."""
    if len(data) > 32:
        raise ValueError("the given vector is longer than 32 bytes")
    b = bytes(reversed(data)) + bytes(32 - len(data))
    return b[:31] + bytes([b[31] & 0x7f])


def scalar_to_be(s: bytes):
    return bytes(reversed(s))


def mimc_hash(preimage: bytes):
    """mimc::mimc_hash (src/mimc_hash/mimc.rs:61-75) -> Scalar bytes (little-endian)."""
    out = _buf(32)
    _chk(lib().bpg_mimc_hash(bytes(preimage), C.c_uint64(len(preimage)), out))
    return out.raw


def scalar_op(op, a, b=None):
    out = _buf(32)
    _chk(lib().bpg_scalar_op(C.c_int32({"add": 0, "sub": 1, "mul": 2, "invert": 3, "reduce": 4, "from_wide": 5}[op]), a, b, out))
    return out.raw


# ------------------------------------------------------------------------------------------------ variables / LCs
class Variable(int):
    """bulletproofs::r1cs::Variable packed as kind << 29 | index."""
    @property
    def kind(self): return int(self) >> 29
    @property
    def index(self): return int(self) & 0x1fffffff
    @staticmethod
    def One(): return Variable(VAR_ONE << 29)


class LinearCombination:
    """List of (Variable, Scalar bytes) terms; From<Variable>, From<Scalar>, +, -, * Scalar as upstream."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def of(x):
        if isinstance(x, LinearCombination):
            return x
        if isinstance(x, Variable):
            return LinearCombination([(x, scalar_from_int(1))])
        if isinstance(x, (bytes, bytearray)):
            return LinearCombination([(Variable.One(), bytes(x))])
        if isinstance(x, int):
            return LinearCombination([(Variable.One(), scalar_from_int(x))])
        raise TypeError(type(x))

    def __add__(self, o): return LinearCombination(self.terms + LinearCombination.of(o).terms)
    def __sub__(self, o): return LinearCombination(self.terms + [(v, scalar_op("sub", bytes(32), c)) for v, c in LinearCombination.of(o).terms])
    def __neg__(self): return LinearCombination([(v, scalar_op("sub", bytes(32), c)) for v, c in self.terms])
    def __mul__(self, s): return LinearCombination([(v, scalar_op("mul", c, s)) for v, c in self.terms])

    def _c(self):
        arr = (_Term * max(len(self.terms), 1))()
        for i, (v, c) in enumerate(self.terms):
            arr[i].var = int(v)
            arr[i].coeff[:] = c
        lc = _LC(C.cast(arr, C.POINTER(_Term)), len(self.terms))
        lc._keep = arr
        return lc


def vars_to_lc(variables):
    return [LinearCombination.of(v) for v in variables]


def _lc_array(lcs):
    cs = [LinearCombination.of(x)._c() for x in lcs]
    arr = (_LC * max(len(cs), 1))(*cs)
    arr._keep = cs
    return arr


# ------------------------------------------------------------------------------------------------ context / generators
class Context:
    """One GPU: PedersenGens (fixed bases) + the BulletproofGens tables resident in HBM."""

    def __init__(self, device=0, profile=None, **config):
        """bpg_ctx_create(device) - the one-shot profile - or, with any of profile="serving" | "oneshot", table_budget_gb, chain_workers,
        chain_lanes, blocking_sync, gens_cache_dir given, bpg_ctx_create_ex with that bpg_config."""
        self._h = C.c_void_p()
        if profile is None and not config:
            _chk(lib().bpg_ctx_create(C.c_int32(device), C.byref(self._h)))
        else:
            cfg = make_config(profile, **config)
            _chk(lib().bpg_ctx_create_ex(C.c_int32(device), C.byref(cfg), C.byref(self._h)))

    def table_bytes(self):
        """bpg_table_bytes: precomputed generator multiples this process holds on the context's device."""
        lib().bpg_table_bytes.restype = C.c_uint64
        return int(lib().bpg_table_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().bpg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pedersen_bases(self):
        a, b = _buf(32), _buf(32)
        _chk(lib().bpg_pedersen_bases(self._h, a, b))
        return a.raw, b.raw

    def gens_ensure(self, capacity):
        _chk(lib().bpg_gens_ensure(self._h, C.c_uint64(capacity)))

    def gens_export(self, first, count):
        g, h = _buf(32 * count), _buf(32 * count)
        _chk(lib().bpg_gens_export(self._h, C.c_uint64(first), C.c_uint64(count), g, h))
        return g.raw[:32 * count], h.raw[:32 * count]

    def pedersen_commit(self, values, blindings):
        k = len(values)
        out = _buf(32 * k)
        _chk(lib().bpg_pedersen_commit(self._h, C.c_uint64(k), b"".join(values), b"".join(blindings), out))
        return [out.raw[32 * i:32 * i + 32] for i in range(k)]

    def msm_gens(self, first, s, t):
        out = _buf(32)
        _chk(lib().bpg_msm_gens(self._h, C.c_uint64(first), C.c_uint64(len(s)), b"".join(s), b"".join(t), out))
        return out.raw

    def test_fe_ops(self, op, a_list, b_list):
        n = len(a_list)
        out = _buf(32 * n)
        _chk(lib().bpg_test_fe_ops(self._h, C.c_int32(op), C.c_uint64(n), b"".join(a_list), b"".join(b_list), out))
        return [out.raw[32 * i:32 * i + 32] for i in range(n)]

    def profile_set(self, mode):
        _chk(lib().bpg_profile_set(self._h, C.c_int32(mode)))

    def _report(self):
        import json
        out = _buf(1 << 16)
        _chk(lib().bpg_profile_report(self._h, out, C.c_uint64(1 << 16)))
        return json.loads(out.value.decode())

    def profile_report(self):
        """bpg_profile_report: {kernel: {count, total_ms, alg_bytes, device_bytes, field_mults}} since the last bpg_profile_set."""
        return {k: v for k, v in self._report().items() if not k.startswith("_")}

    def schedule(self):
        """The "_schedule" object of bpg_profile_report: what the knobs, the profile and the table budget of this context settled on (tail start,
        fold groups, window caps, NAF width and parts in use, epilogue segment, whether the last proof took the shared-device variants)."""
        return self._report()["_schedule"]

    def bench_fe_mul(self, iters=2000):
        r = C.c_double()
        _chk(lib().bpg_bench_fe_mul(self._h, C.c_uint32(iters), C.byref(r)))
        return r.value

    def verify_flat(self, inst: "FlatInstance", transcript_state, commitments, proof, seed=None, flags=0):
        """bpg_r1cs_verify: 0 = accepted, 3 = VERIFICATION_ERROR, 2 = FORMAT_ERROR, 1 = INVALID_GENERATORS_LENGTH."""
        return _verify_flat(self, inst, transcript_state, commitments, proof, seed, flags)

    # ---- PART 1 boundary on flattened instances
    def upload(self, inst: "FlatInstance"):
        h = C.c_void_p()
        cs = inst.cstruct()
        _chk(lib().bpg_r1cs_upload(self._h, C.byref(cs), C.byref(h)))
        return ResidentCircuit(self, h, inst.n, inst.m)

    def blinding_begin(self, transcript_state, v_blinding, rng_seed, max_multipliers):
        """bpg_blinding_begin: start the blinding chain of the next prove on this context (state after every "V" append, m x 32 blinding bytes)."""
        ts = _buf(203); ts.raw = _exact("transcript_state", transcript_state, 203)
        v_blinding = bytes(v_blinding)
        if len(v_blinding) % 32:
            raise ValueError("v_blinding must be a multiple of 32 bytes")
        _chk(lib().bpg_blinding_begin(self._h, ts, C.c_uint64(len(v_blinding) // 32), v_blinding, _seed32(rng_seed), C.c_uint64(max_multipliers)))

    def attach_chain_pool(self, pool, max_streams=2):
        """bpg_ctx_attach_chain_pool: this context's blinding streams are drawn by the shared pool (None: back to its own chain worker)."""
        _chk(lib().bpg_ctx_attach_chain_pool(self._h, pool._h if pool is not None else None, C.c_uint32(max_streams)))
        self._pool = pool                                     # the pool must outlive the attachment

    def set_chain_lanes(self, lanes: int):
        """bpg_ctx_set_chain_lanes: queued blinding streams each chain thread draws in lockstep (1..8; workers * lanes + 1 streams may be alive)."""
        _chk(lib().bpg_ctx_set_chain_lanes(self._h, C.c_uint32(lanes)))

    def set_chain_workers(self, workers: int):
        """bpg_ctx_set_chain_workers: threads that draw queued blinding streams side by side (workers + 1 streams may be alive)."""
        _chk(lib().bpg_ctx_set_chain_workers(self._h, C.c_uint32(workers)))

    def chain_cpu(self):
        """Host core the chain worker of this context last drew a blinding stream on (-1: none yet)."""
        lib().bpg_chain_cpu.restype = C.c_int32
        return int(lib().bpg_chain_cpu(self._h))

    def prove_flat(self, inst: "FlatInstance", transcript_state, v_blinding, rng_seed=None, flags=0):
        ts = _buf(203); ts.raw = _exact("transcript_state", transcript_state, 203)
        v_blinding, rng_seed = _exact("v_blinding", v_blinding, 32 * inst.m), _seed32(rng_seed)
        cap = lib().bpg_proof_size(inst.n, flags)
        out = _buf(cap); ln = C.c_uint64(cap)
        cs = inst.cstruct()
        _chk(lib().bpg_r1cs_prove(self._h, C.byref(cs), ts, C.c_uint64(inst.m), v_blinding, rng_seed, C.c_uint32(flags), out, C.byref(ln)))
        return out.raw[:ln.value], ts.raw[:203]


class ChainPool:
    """bpg_chain_pool_*: host threads that draw the blinding chains of every context attached to them; lanes[k] = chains thread k draws in lockstep."""

    def __init__(self, lanes):
        lanes = [int(x) for x in lanes]
        self._h = C.c_void_p()
        arr = (C.c_uint32 * len(lanes))(*lanes)
        _chk(lib().bpg_chain_pool_create(C.c_uint32(len(lanes)), arr, C.byref(self._h)))
        self.lanes, self.capacity = lanes, sum(lanes)

    def close(self):
        if getattr(self, "_h", None):
            lib().bpg_chain_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _verify_flat(ctx, inst, transcript_state, commitments, proof, seed=None, flags=0):
    ts = _buf(203); ts.raw = _exact("transcript_state", transcript_state, 203)
    seed, commitments, proof = _seed32(seed), _exact("commitments", commitments, 32 * inst.m), bytes(proof)
    cs = inst.cstruct()
    cs.aL = cs.aR = cs.aO = None
    return lib().bpg_r1cs_verify(ctx._h, C.byref(cs), ts, C.c_uint64(inst.m), commitments, proof, C.c_uint64(len(proof)), seed, C.c_uint32(flags))


class _BatchItem(C.Structure):
    _fields_ = [("inst", C.POINTER(R1CSInstance)), ("transcript_state", C.c_void_p), ("m", C.c_uint64), ("v_blinding", C.c_char_p),
                ("rng_seed", C.c_char_p), ("flags", C.c_uint32), ("proof_out", C.c_void_p), ("proof_len", C.POINTER(C.c_uint64))]


class ProverPool:
    """bpg_pool_*: `workers` engine contexts + host threads on one GPU that prove a batch of independent instances concurrently (the
    serial TranscriptRng chain of one proof overlaps the kernels of the others). Same bytes as proving the items one by one."""

    def __init__(self, device: int = 0, workers: int = 8, gens_capacity: int = 0, profile=None, **config):
        self._h = C.c_void_p()
        cfg = make_config(profile, **config) if (profile is not None or config) else None
        _chk(lib().bpg_pool_create_ex(C.c_int32(device), C.c_uint32(workers), C.c_uint64(gens_capacity), C.byref(cfg) if cfg is not None else None, C.byref(self._h)))
        self.workers = workers

    def prove_batch(self, items):
        """items: [(FlatInstance, transcript_state, v_blinding, rng_seed, flags)] -> [(proof bytes, transcript state after)]"""
        n = len(items)
        arr = (_BatchItem * max(n, 1))()
        keep = []
        for k, (inst, state, vb, seed, flags) in enumerate(items):
            cs = inst.cstruct()
            ts = _buf(203); ts.raw = bytes(state)
            cap = lib().bpg_proof_size(inst.n, flags)
            out = _buf(cap); ln = C.c_uint64(cap)
            keep.append((cs, ts, out, ln, vb, seed))
            arr[k].inst = C.pointer(cs); arr[k].transcript_state = C.cast(ts, C.c_void_p); arr[k].m = inst.m
            arr[k].v_blinding = vb; arr[k].rng_seed = seed; arr[k].flags = flags
            arr[k].proof_out = C.cast(out, C.c_void_p); arr[k].proof_len = C.pointer(ln)
        status = (C.c_int32 * max(n, 1))()
        _chk(lib().bpg_pool_prove(self._h, C.c_uint64(n), arr, status))
        return [(k[2].raw[:k[3].value], k[1].raw[:203]) for k in keep]

    def close(self):
        if getattr(self, "_h", None):
            lib().bpg_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentCircuit:
    def __init__(self, ctx, h, n, m):
        self.ctx, self._h, self.n, self.m = ctx, h, n, m

    def prove(self, transcript_state, v_blinding, rng_seed=None, flags=0, timings=False):
        ts = _buf(203); ts.raw = _exact("transcript_state", transcript_state, 203)
        v_blinding, rng_seed = _exact("v_blinding", v_blinding, 32 * self.m), _seed32(rng_seed)
        cap = lib().bpg_proof_size(self.n, flags)
        out = _buf(cap); ln = C.c_uint64(cap)
        tm = Timings()
        _chk(lib().bpg_r1cs_prove_resident(self.ctx._h, self._h, ts, C.c_uint64(self.m), v_blinding, rng_seed, C.c_uint32(flags),
                                           out, C.byref(ln), C.byref(tm) if timings else None))
        return (out.raw[:ln.value], ts.raw[:203], tm.as_dict()) if timings else (out.raw[:ln.value], ts.raw[:203])

    def verify(self, transcript_state, commitments, proof, seed=None, flags=0):
        """bpg_r1cs_verify_resident: 0 = accepted, 3 = VERIFICATION_ERROR, 2 = FORMAT_ERROR, 1 = INVALID_GENERATORS_LENGTH."""
        ts = _buf(203); ts.raw = _exact("transcript_state", transcript_state, 203)
        seed, commitments, proof = _seed32(seed), _exact("commitments", commitments, 32 * self.m), bytes(proof)
        return lib().bpg_r1cs_verify_resident(self.ctx._h, self._h, ts, C.c_uint64(self.m), commitments, proof, C.c_uint64(len(proof)), seed, C.c_uint32(flags))

    def free(self):
        if self._h:
            lib().bpg_r1cs_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FlatInstance:
    """Owned copy of a bpg_r1cs_instance (numpy arrays + bytes); also what tests hand to the oracle."""

    def __init__(self, view: R1CSInstance, v=None, v_blinding=None, commitments=None):
        import numpy as np
        self.n, self.q, self.m, self.nnz, self.ncoef = view.n, view.q, view.m, view.nnz, view.ncoef
        def grab(ptr, nbytes): return C.string_at(ptr, nbytes) if ptr and nbytes else b""
        self.aL, self.aR, self.aO = grab(view.aL, 32 * self.n), grab(view.aR, 32 * self.n), grab(view.aO, 32 * self.n)
        self.row_ptr = np.frombuffer(grab(view.row_ptr, 8 * (self.q + 1)), dtype=np.uint64).copy()
        self.term_var = np.frombuffer(grab(view.term_var, 4 * self.nnz), dtype=np.uint32).copy()
        self.term_coef = np.frombuffer(grab(view.term_coef, 4 * self.nnz), dtype=np.uint32).copy()
        self.coef = grab(view.coef, 32 * self.ncoef)
        self.v, self.v_blinding, self.commitments = v, v_blinding, commitments

    def cstruct(self):
        c = R1CSInstance()
        c.n, c.q, c.m, c.nnz, c.ncoef = self.n, self.q, self.m, self.nnz, self.ncoef
        # one set of C buffers per instance, shared by every struct handed out (a struct must never outlive or invalidate another:
        # a batch holds many structs of the same instance at once)
        src = (self.aL, self.aR, self.aO, self.coef)
        if getattr(self, "_cbufs_src", None) is None or any(a is not b for a, b in zip(self._cbufs_src, src)):
            self._cbufs = [C.create_string_buffer(x, max(len(x), 1)) for x in src]
            self._cbufs_src = src
        c.aL, c.aR, c.aO, c.coef = [C.cast(k, C.c_void_p).value for k in self._cbufs]
        c.row_ptr, c.term_var, c.term_coef = self.row_ptr.ctypes.data, self.term_var.ctypes.data, self.term_coef.ctypes.data
        c._owner = (self, self._cbufs)                       # keeps the buffers alive as long as the struct is
        return c


# ------------------------------------------------------------------------------------------------ transcript / prover / verifier
class Transcript:
    """merlin::Transcript::new(label)."""

    def __init__(self, label: bytes):
        self._h = C.c_void_p()
        self.label = bytes(label)           # a verifier rebuilds its transcript from the same label
        _chk(lib().bpg_transcript_new(self.label, C.c_uint64(len(label)), C.byref(self._h)))

    def append_message(self, label: bytes, msg: bytes):
        _chk(lib().bpg_transcript_append_message(self._h, label, bytes(msg), C.c_uint64(len(msg))))

    def challenge_bytes(self, label: bytes, n: int):
        out = _buf(n)
        _chk(lib().bpg_transcript_challenge_bytes(self._h, label, out, C.c_uint64(n)))
        return out.raw[:n]

    @property
    def state(self):
        out = _buf(203)
        _chk(lib().bpg_transcript_state(self._h, out))
        return out.raw[:203]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().bpg_transcript_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the module globals may be gone
            pass


class Prover:
    """bulletproofs::r1cs::Prover::new(&pc_gens, &mut transcript)."""

    def __init__(self, ctx: Context, transcript: Transcript):
        self.ctx, self.transcript = ctx, transcript
        self._h = C.c_void_p()
        # ctx=None gives an assembly-only prover (no commitments, no prove) - used by the CPU tests of the host logic
        _chk(lib().bpg_prover_new(ctx._h if ctx is not None else None, transcript._h, C.byref(self._h)))

    def commit(self, v: bytes, v_blinding: bytes):
        """Prover::commit(v, v_blinding) -> (CompressedRistretto, Variable)."""
        com, var = _buf(32), C.c_uint32()
        _chk(lib().bpg_prover_commit(self._h, v, v_blinding, com, C.byref(var)))
        return com.raw, Variable(var.value)

    def commit_many(self, vs, blindings):
        k = len(vs)
        coms, vars_ = _buf(32 * k), (C.c_uint32 * max(k, 1))()
        _chk(lib().bpg_prover_commit_many(self._h, C.c_uint64(k), b"".join(vs), b"".join(blindings), coms, vars_))
        return [coms.raw[32 * i:32 * i + 32] for i in range(k)], [Variable(vars_[i]) for i in range(k)]

    def commit_precomputed(self, v: bytes, v_blinding: bytes, commitment: bytes):
        """bpg_prover_commit_precomputed: Prover::commit with the Pedersen commitment supplied by the caller (works without a device context)."""
        var = C.c_uint32()
        _chk(lib().bpg_prover_commit_precomputed(self._h, _exact("v", v, 32), _exact("v_blinding", v_blinding, 32), _exact("commitment", commitment, 32), C.byref(var)))
        return Variable(var.value)

    def test_stub_commitments(self):
        """bpg_test_prover_stub_commitments (TEST HOOK): commitments of this prover become 32 hash bytes of (value, blinding) made on the host - not group
        elements - so that a device-less prover can run a driver's parsing and assembly under sanitizers; prove() stays refused."""
        _chk(lib().bpg_test_prover_stub_commitments(self._h))

    def defer_commitments(self, on: bool = True):
        """Extension (bpg_prover_defer_commitments): while on, commit / commit_many / Gadget.setup register their variables and return zero bytes;
        flush_commitments() computes every pending commitment in one kernel launch and appends them to the transcript in commit order."""
        _chk(lib().bpg_prover_defer_commitments(self._h, C.c_int32(1 if on else 0)))

    def flush_commitments(self):
        _chk(lib().bpg_prover_flush_commitments(self._h))

    def commitment(self, index: int) -> bytes:
        """The commitment of committed variable `index` (commit order), once flushed."""
        out = _buf(32)
        _chk(lib().bpg_prover_commitment(self._h, C.c_uint64(index), out))
        return out.raw

    def multiply(self, left, right):
        out = (C.c_uint32 * 3)()
        l, r = LinearCombination.of(left)._c(), LinearCombination.of(right)._c()
        _chk(lib().bpg_prover_multiply(self._h, C.byref(l), C.byref(r), out))
        return Variable(out[0]), Variable(out[1]), Variable(out[2])

    def allocate_multiplier(self, assignment):
        out = (C.c_uint32 * 3)()
        if assignment is None:
            _chk(lib().bpg_prover_allocate_multiplier(self._h, C.c_int32(0), None, None, out))
        else:
            _chk(lib().bpg_prover_allocate_multiplier(self._h, C.c_int32(1), assignment[0], assignment[1], out))
        return Variable(out[0]), Variable(out[1]), Variable(out[2])

    def allocate(self, assignment):
        out = C.c_uint32()
        _chk(lib().bpg_prover_allocate(self._h, C.c_int32(0 if assignment is None else 1), assignment, C.byref(out)))
        return Variable(out.value)

    def constrain(self, lc):
        c = LinearCombination.of(lc)._c()
        _chk(lib().bpg_prover_constrain(self._h, C.byref(c)))

    def num_constraints(self): return lib().bpg_prover_num_constraints(self._h)
    def get_num_multiplications(self): return lib().bpg_prover_num_multiplications(self._h)
    def num_committed(self): return lib().bpg_prover_num_committed(self._h)

    def instance(self) -> FlatInstance:
        view, v, vb = R1CSInstance(), C.c_void_p(), C.c_void_p()
        _chk(lib().bpg_prover_instance(self._h, C.byref(view), C.byref(v), C.byref(vb)))
        m = view.m
        return FlatInstance(view, v=C.string_at(v, 32 * m) if m else b"", v_blinding=C.string_at(vb, 32 * m) if m else b"")

    def start_blinding(self, rng_seed: bytes = None, max_multipliers: int = 1 << 20):
        """Extension (include/bpg.h bpg_prover_start_blinding): all commitments made - start the serial TranscriptRng chain of the coming
        prove(rng_seed) on a host thread while the constraints are still being assembled. The proof bytes do not change."""
        self._stream_seed = _seed32(rng_seed)      # prove() without an explicit seed continues with this one
        _chk(lib().bpg_prover_start_blinding(self._h, self._stream_seed, C.c_uint64(max_multipliers)))

    def prove(self, bp_gens, rng_seed: bytes = None, flags: int = 0):
        """Prover::prove(&bp_gens) -> R1CSProof::to_bytes(); rng_seed stands in for thread_rng(): 32 fresh random bytes unless given
        (a pinned seed is for tests and benchmarks only)."""
        if rng_seed is None and getattr(self, "_stream_seed", None) is not None:
            rng_seed = self._stream_seed
        rng_seed = _seed32(rng_seed)
        capacity = bp_gens.gens_capacity if isinstance(bp_gens, BulletproofGens) else int(bp_gens)
        cap = lib().bpg_proof_size(self.get_num_multiplications(), flags)
        out = _buf(cap); ln = C.c_uint64(cap)
        _chk(lib().bpg_prover_prove(self._h, C.c_uint64(capacity), rng_seed, C.c_uint32(flags), out, C.byref(ln), None))
        return out.raw[:ln.value]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().bpg_prover_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the module globals may be gone
            pass


class Verifier:
    """bulletproofs::r1cs::Verifier::new(&mut transcript): constraint assembly on the host, verify() / is_valid() on the GPU (bpg_verifier_verify)."""

    def __init__(self, transcript: Transcript):
        self.transcript = transcript
        self._h = C.c_void_p()
        _chk(lib().bpg_verifier_new(transcript._h, C.byref(self._h)))

    def commit(self, com: bytes):
        var = C.c_uint32()
        _chk(lib().bpg_verifier_commit(self._h, com, C.byref(var)))
        return Variable(var.value)

    def get_num_vars(self): return lib().bpg_verifier_num_vars(self._h)

    def verify(self, proof: bytes, ctx: "Context", bp_gens, seed: bytes = None, flags: int = 0):
        """Verifier::verify(&proof, &pc_gens, &bp_gens) on the GPU: returns None, raises BpgError(VERIFICATION_ERROR / FORMAT_ERROR ...)."""
        capacity = bp_gens.gens_capacity if isinstance(bp_gens, BulletproofGens) else int(bp_gens)
        seed, proof = _seed32(seed), bytes(proof)
        _chk(lib().bpg_verifier_verify(self._h, ctx._h, C.c_uint64(capacity), proof, C.c_uint64(len(proof)), seed, C.c_uint32(flags)))

    def is_valid(self, proof, ctx, bp_gens, seed=None, flags=0):
        try:
            self.verify(proof, ctx, bp_gens, seed, flags)
            return True
        except BpgError as e:
            if e.status in (2, 3):
                return False
            raise

    def instance(self) -> FlatInstance:
        view, coms = R1CSInstance(), C.c_void_p()
        _chk(lib().bpg_verifier_instance(self._h, C.byref(view), C.byref(coms)))
        return FlatInstance(view, commitments=C.string_at(coms, 32 * view.m) if view.m else b"")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().bpg_verifier_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the module globals may be gone
            pass


class ConstraintBuffer:
    """ProverBuffer / VerifierBuffer of the reference (src/cs_buffer.rs): records a clause's operations for an OR block."""

    def __init__(self, parent, prover_side: bool):
        first = parent.next_multiplier() if isinstance(parent, ConstraintBuffer) else \
            (parent.get_num_multiplications() if isinstance(parent, Prover) else parent.get_num_vars())
        self.prover_side = prover_side
        self._h = C.c_void_p()
        _chk(lib().bpg_buffer_new(C.c_uint64(first), C.c_int32(1 if prover_side else 0), C.byref(self._h)))

    def rewind(self):
        _chk(lib().bpg_buffer_rewind(self._h))

    def next_multiplier(self):
        lib().bpg_buffer_next_multiplier.restype = C.c_uint64
        return lib().bpg_buffer_next_multiplier(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().bpg_buffer_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the module globals may be gone
            pass


def or_conjunction(main, buffer: ConstraintBuffer):
    """or(main, buffer) (src/or/or_conjunction.rs:4-38): main is a Prover, a Verifier or an enclosing ConstraintBuffer."""
    if isinstance(main, ConstraintBuffer):
        _chk(lib().bpg_or_buffer(main._h, buffer._h))
    elif isinstance(main, Prover):
        _chk(lib().bpg_or_prover(main._h, buffer._h))
    else:
        _chk(lib().bpg_or_verifier(main._h, buffer._h))


class PedersenGens:
    """PedersenGens::default() - the bases live in the Context."""
    def __init__(self, ctx: Context): self.ctx = ctx
    def bases(self): return self.ctx.pedersen_bases()


class BulletproofGens:
    """BulletproofGens::new(gens_capacity, 1): derives the tables on the GPU and keeps them in HBM."""
    def __init__(self, ctx: Context, gens_capacity: int, party_capacity: int = 1):
        if party_capacity != 1:
            raise ValueError("only party_capacity = 1 is used by the reference (src/bin/prover.rs:92)")
        self.ctx, self.gens_capacity = ctx, gens_capacity
        ctx.gens_ensure(gens_capacity)


# ------------------------------------------------------------------------------------------------ commitments.rs
def commit_single(prover: Prover, witness: bytes, blinding: bytes):
    """commitments::commit_single (src/commitments.rs:22-30)."""
    assert len(witness) <= 32, "the provided witness is longer than 32 bytes"
    s = be_to_scalar(witness)
    com, var = prover.commit(s, blinding)
    return s, com, var


def commit(prover: Prover, witness: bytes, blindings):
    """commitments::commit (src/commitments.rs:34-43): splits into 32-byte scalars."""
    scalars = be_to_scalars(witness)
    coms, vars_ = prover.commit_many(scalars, list(blindings)[:len(scalars)])
    return scalars, coms, vars_


def commit_all_single(prover: Prover, witnesses, blindings):
    """commitments::commit_all_single (src/commitments.rs:8-19), batched into one kernel launch."""
    scalars = [be_to_scalar(w) for w in witnesses]
    coms, vars_ = prover.commit_many(scalars, list(blindings)[:len(scalars)])
    return scalars, coms, vars_


def verifier_commit(verifier: Verifier, commitments):
    return [verifier.commit(c) for c in commitments]


# ------------------------------------------------------------------------------------------------ gadgets
class Gadget:
    def __init__(self, handle): self._h = handle

    def setup(self, prover: Prover, witnesses, blindings):
        """Gadget::setup (src/gadget.rs:18-38) -> (commitments, [(scalar, Variable)])."""
        cap = max(8, len(blindings), 2 * len(witnesses) + 2)
        n = C.c_uint64(cap)
        coms, dsc, dvars = _buf(32 * cap), _buf(32 * cap), (C.c_uint32 * cap)()
        _chk(lib().bpg_gadget_setup(self._h, prover._h, b"".join(witnesses), C.c_uint64(len(witnesses)), b"".join(blindings),
                                    C.c_uint64(len(blindings)), coms, dsc, dvars, C.byref(n)))
        k = n.value
        return [coms.raw[32 * i:32 * i + 32] for i in range(k)], [(dsc.raw[32 * i:32 * i + 32], Variable(dvars[i])) for i in range(k)]

    def preprocess(self, witnesses):
        """Gadget::preprocess (src/gadget.rs:8): the derived scalars, for a host that makes the commitments itself."""
        cap = max(8, 2 * len(witnesses) + 2)
        n = C.c_uint64(cap)
        dsc = _buf(32 * cap)
        _chk(lib().bpg_gadget_preprocess(self._h, b"".join(witnesses), C.c_uint64(len(witnesses)), dsc, C.byref(n)))
        return [dsc.raw[32 * i:32 * i + 32] for i in range(n.value)]

    def prove(self, prover, commitment_vars, derived_witnesses):
        """Gadget::prove on a Prover or on a ConstraintBuffer (the dyn ConstraintSystem of the reference)."""
        v = (C.c_uint32 * max(len(commitment_vars), 1))(*[int(x) for x in commitment_vars])
        dv = (C.c_uint32 * max(len(derived_witnesses), 1))(*[int(x[1]) for x in derived_witnesses])
        fn = lib().bpg_gadget_prove_buffered if isinstance(prover, ConstraintBuffer) else lib().bpg_gadget_prove
        _chk(fn(self._h, prover._h, v, C.c_uint64(len(commitment_vars)), b"".join(x[0] for x in derived_witnesses),
                dv, C.c_uint64(len(derived_witnesses))))

    def verify(self, verifier, witnesses, derived):
        v = (C.c_uint32 * max(len(witnesses), 1))(*[int(x) for x in witnesses])
        dv = (C.c_uint32 * max(len(derived), 1))(*[int(x) for x in derived])
        fn = lib().bpg_gadget_verify_buffered if isinstance(verifier, ConstraintBuffer) else lib().bpg_gadget_verify
        _chk(fn(self._h, verifier._h, v, C.c_uint64(len(witnesses)), dv, C.c_uint64(len(derived))))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().bpg_gadget_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the module globals may be gone
            pass


class BoundsCheck(Gadget):
    """BoundsCheck::new(&min, &max) (src/bounds_check/bounds_check_gadget.rs:54-63); big-endian byte vectors."""
    def __init__(self, min_be: bytes, max_be: bytes):
        h = C.c_void_p()
        _chk(lib().bpg_bounds_check_new(bytes(min_be), C.c_uint64(len(min_be)), bytes(max_be), C.c_uint64(len(max_be)), C.byref(h)))
        super().__init__(h)


class MimcHash256(Gadget):
    """MimcHash256::new(image) (src/mimc_hash/mimc_hash_gadget.rs:65-70)."""
    def __init__(self, image):
        h = C.c_void_p()
        lc = LinearCombination.of(image)._c()
        _chk(lib().bpg_mimc_hash256_new(C.byref(lc), C.byref(h)))
        super().__init__(h)


class MerkleTree256(Gadget):
    """MerkleTree256::new(root, instance_vars, witness_vars, pattern) (src/merkle_tree/merkle_tree_gadget.rs:59-73).
    pattern: the tree in the .gadgets syntax with W / I leaves, e.g. "((W W) (I W))"."""
    def __init__(self, root, instance_vars, witness_vars, pattern: str):
        h = C.c_void_p()
        r = LinearCombination.of(root)._c()
        iv, wv = _lc_array(instance_vars), _lc_array(witness_vars)
        _chk(lib().bpg_merkle_tree256_new(C.byref(r), iv, C.c_uint64(len(instance_vars)), wv, C.c_uint64(len(witness_vars)),
                                          pattern.encode(), C.byref(h)))
        super().__init__(h)


class Equality(Gadget):
    """Equality::new(right_hand) (src/equality/equality_gadget.rs:35-40)."""
    def __init__(self, right_hand):
        h = C.c_void_p()
        arr = _lc_array(right_hand)
        _chk(lib().bpg_equality_new(arr, C.c_uint64(len(right_hand)), C.byref(h)))
        super().__init__(h)


class Inequality(Gadget):
    """Inequality::new(right_hand, right_hand_assignment) (src/inequality/inequality_gadget.rs:95-101)."""
    def __init__(self, right_hand, right_hand_assignment=None):
        h = C.c_void_p()
        arr = _lc_array(right_hand)
        ra = b"".join(right_hand_assignment) if right_hand_assignment is not None else None
        if ra == b"":
            ra = bytes(32)      # non-NULL marker for an empty assignment list
        _chk(lib().bpg_inequality_new(arr, C.c_uint64(len(right_hand)), ra, C.byref(h)))
        super().__init__(h)


class LessThan(Gadget):
    """LessThan::new(left, left_assignment, right, right_assignment) (src/less_than/less_than_gadget.rs:69-86)."""
    def __init__(self, left_hand, left_hand_assignment, right_hand, right_hand_assignment):
        h = C.c_void_p()
        l, r = LinearCombination.of(left_hand)._c(), LinearCombination.of(right_hand)._c()
        _chk(lib().bpg_less_than_new(C.byref(l), left_hand_assignment, C.byref(r), right_hand_assignment, C.byref(h)))
        super().__init__(h)


class SetMembership(Gadget):
    """SetMembership::new(value, value_assignment, instance_vars, instance_vars_assignments) (set_membership_gadget.rs:64-77)."""
    def __init__(self, value, value_assignment, instance_vars, instance_vars_assignments):
        h = C.c_void_p()
        v = LinearCombination.of(value)._c()
        arr = _lc_array(instance_vars)
        ia = b"".join(instance_vars_assignments) if instance_vars_assignments else None
        _chk(lib().bpg_set_membership_new(C.byref(v), value_assignment, arr, C.c_uint64(len(instance_vars)), ia, C.byref(h)))
        super().__init__(h)


def range_proof(cs, x, n_bits, x_assignment=None):
    """utils::range_proof (src/utils.rs:5-35) on a Prover (assignment given) or a Verifier (None)."""
    lc = LinearCombination.of(x)._c()
    if isinstance(cs, Prover):
        if x_assignment is None:
            raise BpgError(5, "missing assignment")
        _chk(lib().bpg_range_proof_prove(cs._h, C.byref(lc), C.c_uint32(n_bits), x_assignment))
    else:
        _chk(lib().bpg_range_proof_verify(cs._h, C.byref(lc), C.c_uint32(n_bits)))


def hash_pattern(left, right):
    """hash!(l, r) macro of the reference (merkle_tree_gadget.rs:7-12) on pattern strings."""
    return "(%s %s)" % (left, right)
