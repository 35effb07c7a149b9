"""The product's device arithmetic headers (csrc/hip/fe.cuh, sc.cuh, ge.cuh) compiled for the HOST by
tests/hostcheck and compared with Python big-ints and the oracle.  No GPU needed; the same source is what
hipcc compiles for gfx950."""
import ctypes as C
import hashlib
import pathlib
import subprocess
import pytest
import oracle_lib as O
import pyref as R

HERE = pathlib.Path(__file__).resolve().parent
P, L = R.P, R.L


@pytest.fixture(scope="module")
def hc():
    so = HERE / "hostcheck" / "libhostcheck.so"
    src = HERE / "hostcheck" / "hostcheck.cpp"
    hdrs = list((HERE.parent / "bulletproofs_gadgets_amd" / "csrc" / "hip").glob("*.cuh")) + [HERE.parent / "bulletproofs_gadgets_amd" / "csrc" / "host" / "fe51.hpp"]
    if not so.exists() or any(p.stat().st_mtime > so.stat().st_mtime for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), str(src)])
    return C.CDLL(str(so))


def rnd(tag, i, nbytes=32):
    return hashlib.shake_256(b"%s-%d" % (tag, i)).digest(nbytes)


EDGE = [0, 1, 2, 19, 37, 38, 39, P - 1, P, P + 1, 2 * P - 1, 2 * P, 2 * P + 1, 2**255 - 1, 2**255, 2**256 - 39, 2**256 - 38,
        2**256 - 1, 2**256 - 2**32, 2**224, (2**256 - 1) ^ (2**128 - 1)]


def test_fe_ops_match_bigint(hc):
    vals = EDGE + [int.from_bytes(rnd(b"fe", i), "little") for i in range(60)]
    out = C.create_string_buffer(32)
    ops = {0: lambda a, b: a * b, 1: lambda a, b: a * a, 2: lambda a, b: a + b, 3: lambda a, b: a - b, 4: lambda a, b: -a, 7: lambda a, b: a}
    for i, a in enumerate(vals):
        for b in (vals[(i * 7 + 3) % len(vals)], vals[(i * 13 + 5) % len(vals)], EDGE[i % len(EDGE)]):
            for op, f in ops.items():
                hc.hc_fe_op(op, out, a.to_bytes(32, "little"), b.to_bytes(32, "little"))
                assert int.from_bytes(out.raw, "little") == f(a, b) % P, (op, hex(a), hex(b))
    for a in vals[:30]:
        hc.hc_fe_op(5, out, a.to_bytes(32, "little"), bytes(32))
        assert int.from_bytes(out.raw, "little") == pow(a % P, P - 2, P)
        hc.hc_fe_op(6, out, a.to_bytes(32, "little"), bytes(32))
        assert int.from_bytes(out.raw, "little") == pow(a % P, (P - 5) // 8, P)


def test_fe_chain_weak_forms(hc):
    out = C.create_string_buffer(32)
    for i in range(20):
        a, b = (EDGE[i % len(EDGE)], int.from_bytes(rnd(b"ch", i), "little"))
        hc.hc_fe_chain(out, a.to_bytes(32, "little"), b.to_bytes(32, "little"), 25)
        x, y = a % P, b % P
        for _ in range(25):
            t = (x * y - (x + y)) % P
            x = (y - t) ** 2 % P
            y = (t - x) % P
        assert int.from_bytes(out.raw, "little") == (x + y) % P


def test_sc_ops(hc):
    vals = [0, 1, L - 1, L, L + 1, 2 * L, 2**252, 2**255 - 1, 2**256 - 1] + [int.from_bytes(rnd(b"sc", i), "little") for i in range(50)]
    out = C.create_string_buffer(32)
    for i, a in enumerate(vals):
        hc.hc_sc_from_bytes(out, a.to_bytes(32, "little"))
        assert int.from_bytes(out.raw, "little") == a % L
        b = vals[(i * 5 + 2) % len(vals)]
        for op, f in {0: lambda a, b: a * b, 1: lambda a, b: a + b, 2: lambda a, b: a - b, 3: lambda a, b: -a}.items():
            hc.hc_sc_op(op, out, a.to_bytes(32, "little"), b.to_bytes(32, "little"))
            assert int.from_bytes(out.raw, "little") == f(a, b) % L, (op, hex(a), hex(b))
    for i in range(40):
        w = rnd(b"wide", i, 64) if i else b"\xff" * 64
        hc.hc_sc_from_wide(out, w)
        assert int.from_bytes(out.raw, "little") == int.from_bytes(w, "little") % L
        assert out.raw == O.sc_wide(w)


def test_elligator_compress_and_group_law(hc, golden):
    out = C.create_string_buffer(32)
    for v in golden["one_way_map"]:
        hc.hc_from_uniform(out, bytes.fromhex(v["uniform"]))
        assert out.raw.hex() == v["point"]
    for i in range(12):
        u = rnd(b"uni", i, 64)
        hc.hc_from_uniform(out, u)
        pt = O.from_uniform(u)
        assert out.raw == pt
        k = (int.from_bytes(rnd(b"k", i), "little") % L) if i else 0
        hc.hc_scalarmul_uniform(out, k.to_bytes(32, "little"), u)
        assert out.raw == O.point_mul(k.to_bytes(32, "little"), pt), i


def test_host_fe51_matches_bigint_and_device_forms(hc):
    """host/fe51.hpp (radix 2^51; the product's serial epilogues: Horner recombination of an MSM's window sums and RFC 9496 encoding on the
    host) against Python big-ints on raw device-form inputs, and against the device-side formulas on whole points."""
    vals = EDGE + [int.from_bytes(rnd(b"fe51", i), "little") for i in range(60)]
    out = C.create_string_buffer(32)
    ops = {0: lambda a, b: a * b, 1: lambda a, b: a * a, 2: lambda a, b: a + b, 3: lambda a, b: a - b, 4: lambda a, b: -a, 7: lambda a, b: a}
    for i, a in enumerate(vals):
        for b in (vals[(i * 7 + 3) % len(vals)], vals[(i * 13 + 5) % len(vals)], EDGE[i % len(EDGE)]):
            for op, f in ops.items():
                hc.hc_fe51_op(op, out, a.to_bytes(32, "little"), b.to_bytes(32, "little"))
                assert int.from_bytes(out.raw, "little") == f(a, b) % P, (op, hex(a), hex(b))
    for a in vals[:30]:
        hc.hc_fe51_op(6, out, a.to_bytes(32, "little"), bytes(32))
        assert int.from_bytes(out.raw, "little") == pow(a % P, (P - 5) // 8, P)
    o1, o2 = C.create_string_buffer(32), C.create_string_buffer(32)
    for i in range(16):
        u = rnd(b"uni51", i, 64)
        k = (int.from_bytes(rnd(b"k51", i), "little") % L) if i else 0          # k = 0: the identity, encoded as 32 zero bytes
        assert hc.hc_fe51_point_check(o1, o2, k.to_bytes(32, "little"), u, 1 + 7 * i) == 1, i
        assert o1.raw == o2.raw == O.point_mul(k.to_bytes(32, "little"), O.from_uniform(u))
