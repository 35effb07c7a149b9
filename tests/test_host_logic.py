"""CPU tests of the product's host side (C++ inside libbpg_hip.so, no GPU needed): C-ABI exports, host scalar
arithmetic, Merlin, MiMC, conversions and gadget assembly.  Circuits assembled by the product are handed to the
ORACLE (tests/oracle_lib.py) for satisfaction / prove / verify checks."""
import hashlib
import pathlib
import re
import pytest
import bulletproofs_gadgets_amd as bpg
import oracle_lib as O
import pyref as R

H = bytes.fromhex
sc = lambda x: (x % R.L).to_bytes(32, "little")


def test_library_exports_every_declared_symbol():
    hdr = (O.ROOT / "include" / "bpg.h").read_text()
    names = set(re.findall(r"\b(bpg_[a-z0-9_]+)\s*\(", hdr)) - {"bpg_status"}
    assert len(names) > 40
    lib = bpg.lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_fails_loudly():
    # on the GPU box this creates a context; without a GPU it must raise DEVICE_ERROR (no CPU fallback)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bpg.BpgError) as e:
        bpg.Context(0)
    assert e.value.status == 7
    t = bpg.Transcript(b"x")
    p = bpg.Prover(None, t)
    with pytest.raises(bpg.BpgError) as e:
        p.commit(sc(1), sc(2))
    assert e.value.status == 7


def test_host_scalar_semantics():
    vals = [0, 1, R.L - 1, R.L, R.L + 5, 2**255 - 1, 2**256 - 1] + [int.from_bytes(hashlib.sha256(b"%d" % i).digest(), "little") for i in range(40)]
    for i, a in enumerate(vals):
        b = vals[(3 * i + 1) % len(vals)]
        A, B = a.to_bytes(32, "little"), b.to_bytes(32, "little")
        assert bpg.scalar_op("add", A, B) == sc(a + b)
        assert bpg.scalar_op("sub", A, B) == sc(a - b)
        assert bpg.scalar_op("mul", A, B) == sc(a * b)
        assert bpg.scalar_op("reduce", A) == sc(a)
        if a % R.L:
            assert bpg.scalar_op("invert", A) == sc(pow(a % R.L, R.L - 2, R.L))
    for i in range(20):
        w = hashlib.sha512(b"w%d" % i).digest() if i else b"\xff" * 64
        assert bpg.scalar_op("from_wide", w) == O.sc_wide(w)


def test_host_merlin_matches_published_vector_and_oracle():
    t = bpg.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    tp, to = bpg.Transcript(b"long"), O.Transcript(b"long")
    for k in range(5):
        msg = bytes(range(256)) * (k + 1)
        tp.append_message(b"blob", msg); to.append(b"blob", msg)
        assert tp.challenge_bytes(b"c", 100 + 90 * k) == to.challenge(b"c", 100 + 90 * k)
        assert tp.state == to.state


def test_conversions_reference_unit_tests():
    # reference src/conversions.rs:100-150
    B1 = H("7b2460be180544cd18e3e7e27330cec9517a314acbd4a011d273a59b480c1e00")
    B2 = H("7b987cf97a9f1bd5492347d6f4e550ae2949a513de92fe5065350ebcd51db604")
    assert bpg.be_to_scalar(B1) == bytes(reversed(B1))
    s = bpg.be_to_scalars(B1 + B2)
    assert s[0] == bytes(reversed(B2)) and s[1] == bytes(reversed(B1))
    assert bpg.scalar_to_be(bpg.be_to_scalar(B1)) == B1
    # from_bits: bit 255 cleared, no reduction
    assert bpg.be_to_scalars(b"\xff" * 32)[0] == b"\xff" * 31 + b"\x7f"
    assert len(bpg.be_to_scalars(b"\x01" * 33)) == 2


def test_mimc_hash_kats(golden):
    m = golden["mimc"]
    be = lambda b: bytes(reversed(b)).hex()
    assert be(bpg.mimc_hash(H(m["kat1_in"]))) == m["kat1_be"]
    assert be(bpg.mimc_hash(H(m["kat2_in"]))) == m["kat2_be"]
    assert be(bpg.mimc_hash(H(m["kat3_in"]))) == m["kat3_be"]
    assert be(bpg.mimc_hash(b"John")) == m["john_be"]
    for pre in [bytes([0xc4]) + bytes(range(31)), bytes([0x80]) + bytes(31), b"\xff" * 64, b"\x00" * 5 + b"\x01", b"a" * 2130]:
        assert bpg.mimc_hash(pre) == O.mimc_hash(pre)


def to_oracle(inst):
    return O.FlatCircuit(inst.n, inst.m, inst.aL or None, inst.aR or None, inst.aO or None, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)


def test_range_proof_assembly_and_oracle_roundtrip():
    # reference src/utils.rs:46-90 (test_range_proof_1 ok with n = 56, test_range_proof_2 must fail with n = 48)
    x = bpg.be_to_scalar(H("0522a64d7b931e"))
    for nbits, ok in ((56, True), (48, False)):
        tp = bpg.Transcript(b"RangeProof")
        p = bpg.Prover(None, tp)
        bpg.range_proof(p, x, nbits, x)
        assert p.get_num_multiplications() == nbits and p.num_constraints() == 2 * nbits + 1
        pi = p.instance()
        assert O.satisfied(to_oracle(pi), b"") == ok
        tv = bpg.Transcript(b"RangeProof")
        v = bpg.Verifier(tv)
        bpg.range_proof(v, x, nbits)
        vi = v.instance()
        assert (vi.n, vi.q) == (nbits, 2 * nbits + 1)
        assert bytes(vi.term_var) == bytes(pi.term_var) and bytes(vi.row_ptr) == bytes(pi.row_ptr)
        gens = O.Gens(64)
        rc, proof, _ = O.prove(gens, tp.state, to_oracle(pi), b"", bytes(32), O.FLAG_FAST_MSM)
        assert rc == 0
        assert (O.verify(gens, tv.state, to_oracle(vi), b"", proof) == 0) == ok


def test_merkle_assembly_sizes_and_witness_synthesis(golden):
    # all-instance tree: no commitments needed, so the prover-side synthesis runs without a GPU
    m = golden["mimc"]
    leaf = bpg.be_to_scalar(H(m["leaf512_be"]))
    root4 = bpg.be_to_scalar(H(m["levels512_be"][1]))        # 4 equal leaves -> level-2 digest (merkle_tree_gadget.rs:479-483)
    tp = bpg.Transcript(b"MerkleTree")
    p = bpg.Prover(None, tp)
    g = bpg.MerkleTree256(root4, [leaf] * 4, [], "((I I) (I I))")
    g.prove(p, [], [])
    assert p.get_num_multiplications() == 3 * 1944 and p.num_constraints() == 3 * 3888 + 1
    inst = p.instance()
    assert O.satisfied(to_oracle(inst), b"")
    # wrong root -> unsatisfied
    p2 = bpg.Prover(None, bpg.Transcript(b"MerkleTree"))
    bpg.MerkleTree256(leaf, [leaf] * 4, [], "((I I) (I I))").prove(p2, [], [])
    assert not O.satisfied(to_oracle(p2.instance()), b"")
    # pattern errors surface as INVALID_ARGUMENT, like the reference's assert
    with pytest.raises(bpg.BpgError):
        bpg.MerkleTree256(root4, [leaf] * 3, [], "((I I) (I I))").prove(bpg.Prover(None, bpg.Transcript(b"x")), [], [])
    with pytest.raises(bpg.BpgError):
        bpg.MerkleTree256(root4, [], [], "((I I) (I")
    # nesting is bounded (Pattern::MAX_DEPTH = 64: parser, assembly and destructor recurse once per level, and the text comes from a file): a pattern of
    # twenty thousand opening brackets is INVALID_ARGUMENT, not a stack overflow across the C ABI
    for bad in ("(" * 20000 + "I", "(" * 70 + "I" + " I)" * 70):
        with pytest.raises(bpg.BpgError):
            bpg.MerkleTree256(root4, [leaf] * 80, [], bad)
    bpg.MerkleTree256(root4, [leaf] * 61, [], "(" * 60 + "I" + " I)" * 60)       # sixty levels are a tree


def test_verifier_side_gadget_sizes():
    # BoundsCheck over 8-byte bounds: 2*64 multipliers, 1 + 2*(2*64+1) constraints (SURVEY.md section 8 cfg 2)
    tv = bpg.Transcript(b"BoundsCheck")
    v = bpg.Verifier(tv)
    w = v.commit(bytes(32))
    d = [v.commit(bytes(32)), v.commit(bytes(32))]
    bpg.BoundsCheck(bytes(8), b"\xff" * 8).verify(v, [w], d)
    i = v.instance()
    assert (i.n, i.q, i.m) == (128, 259, 3)
    # MimcHash256 over one block, happy padding case: 972 multipliers, 1946 constraints (src/or/or_conjunction.rs:85)
    v = bpg.Verifier(bpg.Transcript(b"MiMCHash"))
    image, pre = v.commit(bytes(32)), v.commit(bytes(32))
    d = [v.commit(bytes(32)), v.commit(bytes(32))]
    bpg.MimcHash256(image).verify(v, [pre], d)
    i = v.instance()
    assert (i.n, i.q, i.m) == (972, 1946, 4)


def test_keccak_vector_and_scalar_implementations_agree():
    import ctypes as C
    impl, ns = C.c_int32(), C.c_double()
    for seed in (1, 2, 0xdeadbeef):
        rc = bpg.lib().bpg_keccak_selftest(C.c_uint64(seed), C.c_uint32(3000), C.byref(impl), None)
        assert rc == 0, bpg.lib().bpg_last_error()
    assert impl.value in (0, 1, 2)


def test_bulk_rng_draws_match_generic_strobe_path_and_oracle():
    """The prover draws s_L, s_R through an in-register bulk path (merlin.hpp rng_draws64); it must give the bytes of
    meta_ad(le32(64)) + prf(64) per draw, from any starting position, and reduce to the oracle's Scalar::random values."""
    import ctypes as C
    import oracle_lib as O
    t = bpg.Transcript(b"rng-bulk")
    t.append_message(b"x", b"y" * 37)
    vb = b"".join(bytes([i + 1]) + bytes(31) for i in range(3))
    seed = bytes(range(32))
    count = 300
    outs = []
    for skip, bulk in ((0, 0), (0, 1), (5, 1), (5, 0)):
        out = C.create_string_buffer(64 * count)
        rc = bpg.lib().bpg_rng_draws(t.state, C.c_uint64(3), vb, seed, C.c_uint64(skip), C.c_uint64(count), C.c_int32(bulk), out)
        assert rc == 0, bpg.lib().bpg_last_error()
        outs.append(out.raw)
    assert outs[0] == outs[1] and outs[2] == outs[3] and outs[0][64 * 5:] == outs[2][:64 * (count - 5)]
    want = O.rng_scalars(t.state, vb, seed, 8)
    got = b"".join(bpg.scalar_op("from_wide", outs[1][64 * i:64 * i + 64]) for i in range(8))
    assert got == b"".join(want)


def test_lockstep_rng_draws_match_the_single_generator():
    """Eight TranscriptRng generators of eight different proofs drawn in lockstep (one sponge per 64-bit lane of ZMM registers, merlin.hpp
    strobe_rng_bulk64_x8; falls back to one after the other without AVX-512): every lane gives the bytes of its own generic STROBE path, from
    different starting positions, for any number of lanes."""
    import ctypes as C
    t = bpg.Transcript(b"rng-lanes")
    t.append_message(b"x", b"z" * 11)
    vb = b"".join(bytes([i + 3]) + bytes(31) for i in range(2))
    count = 700
    for lanes in (1, 2, 5, 8):
        seeds = b"".join(bytes([17 * v + 1]) * 32 for v in range(lanes))
        skips = [(3 * v) % 4 for v in range(lanes)]                       # some lanes start at the steady position, some do not
        out = C.create_string_buffer(lanes * 64 * count)
        rc = bpg.lib().bpg_rng_draws_multi(t.state, C.c_uint64(2), vb, C.c_uint32(lanes), seeds, (C.c_uint64 * lanes)(*skips), C.c_uint64(count), out)
        assert rc == 0, bpg.lib().bpg_last_error()
        for v in range(lanes):
            ref = C.create_string_buffer(64 * count)
            rc = bpg.lib().bpg_rng_draws(t.state, C.c_uint64(2), vb, seeds[32 * v:32 * v + 32], C.c_uint64(skips[v]), C.c_uint64(count), C.c_int32(0), ref)
            assert rc == 0
            assert out.raw[v * 64 * count:(v + 1) * 64 * count] == ref.raw, (lanes, v)


def test_or_conjunction_assembly_counts_on_the_verifier_side():
    """src/or/or_conjunction.rs:4-38 with the recording buffer of src/cs_buffer.rs: the clauses' multipliers are replayed into the
    parent, then one product chain per element of the Cartesian product of the clauses' explicit constraints.
    BOUND with 1-byte bounds = 2 x 8-bit range proofs: 16 allocate_multiplier calls, 35 explicit constraints."""
    def bound_clause(cs, coms):
        bpg.BoundsCheck(bytes([10]), bytes([100])).verify(cs, [coms[0]], coms[1:3])

    v0 = bpg.Verifier(bpg.Transcript(b"or"))
    c0 = [v0.commit(bytes(32)) for _ in range(3)]
    bound_clause(v0, c0)
    i0 = v0.instance()
    assert (i0.n, i0.q) == (16, 35)

    for k in (2, 3):
        v = bpg.Verifier(bpg.Transcript(b"or"))
        coms = [v.commit(bytes(32)) for _ in range(3 * k)]
        buf = bpg.ConstraintBuffer(v, False)
        for j in range(k):
            bound_clause(buf, coms[3 * j:3 * j + 3])
            buf.rewind()
        assert buf.next_multiplier() == 16 * k
        bpg.or_conjunction(v, buf)
        i = v.instance()
        products = 35 ** k
        assert i.n == 16 * k + products * (k - 1)
        assert i.q == products + 2 * products * (k - 1)          # one constraint per product + 2 per multiply()
    # nested: OR( A, OR(B, C) ) - the inner OR replays into the outer clause's buffer
    v = bpg.Verifier(bpg.Transcript(b"or"))
    coms = [v.commit(bytes(32)) for _ in range(9)]
    outer = bpg.ConstraintBuffer(v, False)
    bound_clause(outer, coms[0:3]); outer.rewind()
    inner = bpg.ConstraintBuffer(outer, False)
    bound_clause(inner, coms[3:6]); inner.rewind()
    bound_clause(inner, coms[6:9]); inner.rewind()
    bpg.or_conjunction(outer, inner); outer.rewind()
    bpg.or_conjunction(v, outer)
    i = v.instance()
    inner_n, inner_c = 32 + 35 * 35, 35 * 35                     # the inner OR as a clause: multipliers, explicit constraints
    assert i.n == 16 + inner_n + 35 * inner_c
    # an OR over no clauses adds nothing
    v = bpg.Verifier(bpg.Transcript(b"or"))
    bpg.or_conjunction(v, bpg.ConstraintBuffer(v, False))
    assert (v.instance().n, v.instance().q) == (0, 0)


def test_blinding_stream_needs_a_device_context():
    # the speculative TranscriptRng stream lives in the engine: an assembly-only prover refuses it loudly, a null context is an argument error
    p = bpg.Prover(None, bpg.Transcript(b"x"))
    with pytest.raises(bpg.BpgError) as e:
        p.start_blinding(bytes(32), 1024)
    assert e.value.status == 7
    import ctypes as C
    rc = bpg.lib().bpg_blinding_begin(None, bytes(203), C.c_uint64(0), None, bytes(32), C.c_uint64(16))
    assert rc != 0 and b"" != bpg.lib().bpg_last_error()


def test_bench_launcher_refuses_a_world_that_is_not_gpus(tmp_path):
    """bench.py --gpus N is one rank of N under torchrun and the launcher of N ranks otherwise; a mismatch must fail loudly, not run 1 rank."""
    import os, subprocess, sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    # without a GPU the product refuses to run at all (no CPU path), also through the launcher
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an AMD GPU" in r.stderr


def test_python_binding_checks_buffer_lengths_and_defaults_to_fresh_seeds():
    import bulletproofs_gadgets_amd as bpg
    assert bpg._seed32(None) != bpg._seed32(None) and len(bpg._seed32(None)) == 32
    assert bpg._seed32(bytes(32)) == bytes(32)
    with pytest.raises(ValueError):
        bpg._seed32(b"short")
    with pytest.raises(ValueError):
        bpg._exact("v_blinding", bytes(31), 32)
    import inspect
    for fn in (bpg.Prover.prove, bpg.Prover.start_blinding, bpg.Verifier.verify, bpg.Verifier.is_valid, bpg.ResidentCircuit.prove, bpg.ResidentCircuit.verify):
        params = inspect.signature(fn).parameters
        name = "rng_seed" if "rng_seed" in params else "seed"
        assert params[name].default is None, fn.__qualname__


def test_profile_tooling_on_synthetic_counter_files(tmp_path):
    """tools/calib/fetch_factor.py and tools/pmc_summarize.py on hand-made rocprofv3 CSVs: the FETCH_SIZE factor is known bytes / counter bytes;
    HBM bytes are FETCH_SIZE x 2 + WRITE_SIZE for EVERY kernel, gather kernels carry the calibrated useful-bytes figure in a field of its own, and
    the result carries the hash of the kernel sources (bench.py quotes `traffic` only when that hash is the one of the sources it runs); with the
    JSON line of a profiled --in-flight-only run the summary also gives the bytes per proof of the concurrent mix."""
    import json, subprocess, sys
    root = pathlib.Path(__file__).resolve().parent.parent
    known = {"table_bytes": 1000, "rows_gathered": 10, "known_bytes": {"k_calib_stream": 2048.0, "k_calib_gather<6>": 960.0, "k_calib_gather<8>": 960.0},
             "ms": {}, "GBps": {}}
    (tmp_path / "known.json").write_text(json.dumps(known) + "\n")
    hdr = "Kernel_Name,Counter_Name,Counter_Value\n"
    (tmp_path / "calib.csv").write_text(hdr + 'k_calib_stream(uint4 const*),FETCH_SIZE,1.0\n"void k_calib_gather<6>(uint4 const*)",FETCH_SIZE,1.0\n'
                                        '"void k_calib_gather<8>(uint4 const*)",FETCH_SIZE,0.5\n')
    subprocess.check_call([sys.executable, str(root / "tools/calib/fetch_factor.py"), str(tmp_path / "known.json"), str(tmp_path / "calib.csv"), str(tmp_path / "calib.json")],
                          stdout=subprocess.DEVNULL)
    cal = json.loads((tmp_path / "calib.json").read_text())["summary"]
    assert cal["stream_factor"] == 2.0 and cal["gather96_factor"] == 960.0 / 1024.0
    (tmp_path / "fetch.csv").write_text(hdr + "bpg::k_bucket_chunks(bpg::MsmSegs),FETCH_SIZE,100\nbpg::k_bucket_chunks(bpg::MsmSegs),FETCH_SIZE,300\nbpg::k_poly_t(x),FETCH_SIZE,10\n")
    (tmp_path / "write.csv").write_text(hdr + "bpg::k_bucket_chunks(bpg::MsmSegs),WRITE_SIZE,4\nbpg::k_bucket_chunks(bpg::MsmSegs),WRITE_SIZE,4\nbpg::k_poly_t(x),WRITE_SIZE,2\n")
    subprocess.check_call([sys.executable, str(root / "tools/pmc_summarize.py"), str(tmp_path / "fetch.csv"), str(tmp_path / "write.csv"), str(tmp_path / "calib.json"),
                           str(tmp_path / "traffic.json")], stdout=subprocess.DEVNULL)
    t = json.loads((tmp_path / "traffic.json").read_text())
    sys.path.insert(0, str(root))
    import bench
    assert t["_meta"]["source_hash"] == bench.source_hash() and len(t["_meta"]["source_hash"]) == 16
    sweep, poly = t["k_bucket_chunks"], t["k_poly_t"]
    assert sweep["launches"] == 2 and sweep["access"] == "gather" and poly["access"] == "stream" and "useful_fetch_bytes_per_launch" not in poly
    assert abs(sweep["hbm_bytes_per_launch"] - (200 * 1024 * 2.0 + 4 * 1024)) < 1e-6                 # raw counter x 2, never the calibrated factor
    assert abs(sweep["useful_fetch_bytes_per_launch"] - 200 * 1024 * 960.0 / 1024.0) < 1e-6
    assert abs(poly["hbm_bytes_per_launch"] - (10 * 1024 * 2.0 + 2 * 1024)) < 1e-6
    assert "k_fold_points_wnaf" in t["_meta"]["gather_kernels"] and "k_tt_round8" in t["_meta"]["gather_kernels"]
    (tmp_path / "line.json").write_text("noise\n" + json.dumps({"in_flight": {"proofs": 48, "proofs_in_flight": 6}}) + "\n")
    subprocess.check_call([sys.executable, str(root / "tools/pmc_summarize.py"), str(tmp_path / "fetch.csv"), str(tmp_path / "write.csv"), str(tmp_path / "calib.json"),
                           str(tmp_path / "thr.json"), str(tmp_path / "line.json")], stdout=subprocess.DEVNULL)
    m = json.loads((tmp_path / "thr.json").read_text())["_meta"]
    assert m["proofs_profiled"] == 48 + 6 + 1 and abs(m["hbm_bytes_per_proof"] - (2 * (200 * 1024 * 2 + 4 * 1024) + 10 * 1024 * 2 + 2 * 1024) / 55.0) < 1e-6
    assert bench._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}


def test_host_thread_planning_on_synthetic_topologies():
    """bench.py sizes the chain threads of a rank for the cores it really has (cores_for_rank) and deals the first chains of all its proving
    streams to them (plan_chain_pool): every chain thread gets a core to itself, two cores stay free, lockstep threads only where cores run out."""
    import bench
    # an 8-GPU host, 2 NUMA nodes of 64 cores, 4 ranks on each: 16 cores per rank -> 13 single-lane threads + one with 7 lanes for 20 streams
    assert bench.cores_for_rank(64, 4) == 16
    lanes = bench.plan_chain_pool(16, 20)
    assert lanes == [1] * 13 + [7] and len(lanes) <= 16 - 2
    # the same host under a cgroup quota of 64 CPUs for the whole job: 8 per rank
    assert bench.cores_for_rank(64, 4, quota=32) == 8
    lanes = bench.plan_chain_pool(8, 20)
    assert sum(lanes) == 20 and len(lanes) == 6 and max(lanes) <= 8 and lanes[:4] == [1] * 4
    # 8 ranks on one 32-core node: 4 cores per rank, two chain threads with 8 lanes each; four chains of twenty wait for the second wave
    assert bench.cores_for_rank(32, 8) == 4
    assert bench.plan_chain_pool(4, 20) == [8, 8]
    # a one-GPU box that shows 256 CPUs but is throttled to 16
    assert bench.cores_for_rank(128, 1, quota=16) == 16
    # plenty of cores: one lane per stream, never more threads than streams
    assert bench.plan_chain_pool(64, 20) == [1] * 20 and bench.plan_chain_pool(64, 3) == [1] * 3
    # the 8-GPU node of BASELINE config 5 with 128 cores in all (profiles/r04_appetite.txt: which row of the appetite table it can host): 16 cores per
    # rank -> 14 chain threads per rank, 112 on the node - the 20-stream row of one GPU, on every GPU at once; with 10 streams per rank every chain has
    # a core to itself (10 threads per rank, 80 on the node), with 6 streams 48
    assert bench.cores_for_rank(64, 4) * 8 == 128
    assert [len(bench.plan_chain_pool(16, n)) for n in (20, 10, 6)] == [14, 10, 6]
    assert [sum(bench.plan_chain_pool(16, n)) for n in (20, 10, 6)] == [20, 10, 6]
    assert 8 * len(bench.plan_chain_pool(16, 20)) == 112 and 8 * (len(bench.plan_chain_pool(16, 20)) + 2) <= 128
    # `bench.py --gpus 8 --plan-only` (no GPU call): the same planner over the whole node, with the memory model of the engine's allocations - eight ranks of
    # twenty streams fit a 288 GB card each and their pinned slabs fit the host RAM the planner is told about, under both profiles
    for profile in ("serving", "oneshot"):
        plan = bench.plan_node(8, 20, 20, profile, bench.synthetic_topology(8, numa_nodes=2, cores_per_node=64, host_ram_GB=1536.0))
        assert plan["fits"] == {"hbm_per_card": True, "pinned_host": True, "a_core_per_chain_thread": True}
        assert [r["numa_node"] for r in plan["ranks"]] == [0] * 4 + [1] * 4 and all(r["chain_pool_lanes"] == [1] * 13 + [7] for r in plan["ranks"])
        assert all(r["hbm_expected_GB"] < 288 for r in plan["ranks"]) and plan["pinned_host_total_GB"] < 0.5 * 1536
    one, srv = bench.memory_model(993384, 1 << 20, 1986769, 512, "oneshot", 20), bench.memory_model(993384, 1 << 20, 1986769, 512, "serving", 20)
    assert one["hbm_GB"] <= 40.0 < srv["hbm_GB"] < 100.0 and abs((srv["hbm_GB"] - one["hbm_GB"]) - 240 * 2 * (1 << 20) * 96 / 1e9) < 0.01
    # a host with 256 GB of RAM and a 64-CPU quota: the pinned slabs of 8 x 20 streams (43.5 GB) still fit, but a chain thread no longer gets a core
    tight = bench.plan_node(8, 20, 20, "serving", bench.synthetic_topology(8, cores_per_node=64, host_ram_GB=256.0, cpu_quota=64))
    assert tight["fits"]["pinned_host"] and all(r["cores_for_rank"] == 8 and sum(r["chain_pool_lanes"]) == 20 for r in tight["ranks"])
    small = bench.plan_node(8, 20, 20, "serving", bench.synthetic_topology(8, host_ram_GB=64.0))
    assert not small["fits"]["pinned_host"]
    for cores in range(1, 40):
        for n in (1, 2, 3, 8, 14, 20, 33):
            lanes = bench.plan_chain_pool(cores, n)
            assert lanes and all(1 <= x <= 8 for x in lanes) and len(lanes) <= max(1, cores - 2) and sum(lanes) <= n
            assert sum(lanes) == n or sum(lanes) >= 8 * max(1, cores - 2) - 7      # short only when the cores cannot carry n chains at all


def test_config_struct_matches_the_header_and_pool_api_without_a_gpu(tmp_path):
    """bpg_config as the Python binding lays it out == as a C compiler lays out include/bpg.h (size and every offset); the chain pool and the
    device count are usable without a device (threads idle, count 0), bad arguments are refused."""
    import ctypes as C
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bpg.h"\nint main(void) { printf("%zu %u\\n", sizeof(bpg_timings), BPG_ABI_VERSION); printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(bpg_config), '
                   'offsetof(bpg_config, struct_size), offsetof(bpg_config, profile), offsetof(bpg_config, table_budget_gb), offsetof(bpg_config, chain_workers), '
                   'offsetof(bpg_config, chain_lanes), offsetof(bpg_config, blocking_sync), offsetof(bpg_config, gens_cache_dir)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", str(O.ROOT / "include"), "-o", str(exe), str(src)])
    want = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    # the caller-owned structs without a size field are frozen: bpg_timings is nine doubles, here and in the binding; the library says which ABI it speaks
    assert want[0] == 72 == C.sizeof(bpg.Timings)
    bpg.lib().bpg_abi_version.restype = C.c_uint32
    assert bpg.lib().bpg_abi_version() == want[1] >= 5
    want = want[2:]
    got = [C.sizeof(bpg.Config)] + [getattr(bpg.Config, f).offset for f, _ in bpg.Config._fields_]
    assert got == want, (got, want)
    cfg = bpg.make_config("serving", table_budget_gb=12.5, chain_workers=3, chain_lanes=4, blocking_sync=True, gens_cache_dir="/tmp/x")
    assert (cfg.struct_size, cfg.profile, cfg.table_budget_gb, cfg.chain_workers, cfg.chain_lanes, cfg.blocking_sync, cfg.gens_cache_dir) == \
        (want[0], 2, 12.5, 3, 4, 1, b"/tmp/x")
    lib = bpg.lib()
    lib.bpg_device_count.restype = C.c_int32
    import torch
    assert lib.bpg_device_count() == torch.cuda.device_count()
    pool = bpg.ChainPool([1, 1, 8])                          # three idle threads
    assert pool.capacity == 10
    pool.close()
    for bad in ([], [0], [9], [1] * 257):
        with pytest.raises(bpg.BpgError):
            bpg.ChainPool(bad)
    if not torch.cuda.is_available():
        with pytest.raises(bpg.BpgError) as e:
            bpg.Context(0, profile="serving", table_budget_gb=1.0)
        assert e.value.status == 7                          # no device: DEVICE_ERROR, also through bpg_ctx_create_ex
    with pytest.raises(bpg.BpgError) as e:
        bpg.Context(0, profile=2, chain_lanes=9)
    assert e.value.status in (4, 7)


@pytest.mark.parametrize("var,val", [("BPG_MERGE", "3"), ("BPG_MERGE", "yes"), ("BPG_RSEG", "0"), ("BPG_RSEG", "3"), ("BPG_RSEG", "abc"), ("BPG_RSEG", "2048"), ("BPG_RSEG", "8x"), ("BPG_LGCH", "1"), ("BPG_LGCH", "zz"),
                                     ("BPG_SWEEP_RESIDENT", "0"), ("BPG_MSM_CMAX", "17"), ("BPG_MSM_CMIN", ""), ("BPG_FOLD_GROUP", "6"), ("BPG_FOLD_WNAF", "2"),
                                     ("BPG_FOLD_PARTS", "3"), ("BPG_TT_LG", "-1"), ("BPG_TABLE_GB", "-4"), ("BPG_FOLD_TABLE_GB", "nan"), ("BPG_CHAIN_LANES", "9"),
                                     ("BPG_SYNC_BLOCKING", "2"), ("BPG_FOLD_ADAPT", "3"), ("BPG_WINDOW_QUAD", "2"), ("BPG_WINDOW_QUAD_BLOCKS", "0"), ("BPG_FOLD_QUAD_W", "on"), ("BPG_FOLD_REG_W", "-1")])
def test_bad_environment_knobs_are_refused_at_context_creation(var, val, monkeypatch):
    """Every BPG_* knob is read once, when the context is created, BEFORE the device is touched: a value that does not parse or is out of range is
    BPG_ERR_INVALID_ARGUMENT from bpg_ctx_create - with or without a GPU - never a silent default and never a fault on the hot path (round 3 read
    BPG_RSEG with atoi on every MSM call and divided by it: SIGFPE across the C ABI).  An EMPTY variable counts as unset."""
    monkeypatch.setenv(var, val)
    import torch
    if val == "":
        if torch.cuda.is_available():
            bpg.Context(0).close()
        else:
            with pytest.raises(bpg.BpgError) as e:
                bpg.Context(0)
            assert e.value.status == 7
        return
    with pytest.raises(bpg.BpgError) as e:
        bpg.Context(0)
    assert e.value.status == 4, (var, val, e.value.status, str(e.value))
    assert var in str(e.value)


def test_stub_commitment_hook_is_device_less_and_never_proves():
    """bpg_test_prover_stub_commitments (the hook behind tests/hostcheck/fuzz_cli): a prover WITHOUT a device context refuses to commit; with the hook
    its commitments are 32 deterministic hash bytes (same value and blinding -> same bytes, in the per-call and in the deferred form), the variables
    and the transcript move as usual, a gadget's setup works - and prove() still fails with BPG_ERR_DEVICE: the hook can never produce a proof."""
    def build(deferred):
        t = bpg.Transcript(b"stub")
        p = bpg.Prover(None, t)
        with pytest.raises(bpg.BpgError) as e:
            p.commit(bpg.be_to_scalar(b"\x05"), bytes(32))
        assert e.value.status == 7
        p.test_stub_commitments()
        if deferred:
            p.defer_commitments(True)
        c1, v1 = p.commit(bpg.be_to_scalar(b"\x05"), bytes([1]) + bytes(31))
        c2, v2 = p.commit(bpg.be_to_scalar(b"\x06"), bytes([1]) + bytes(31))
        bpg.range_proof(p, v1, 8, bpg.be_to_scalar(b"\x05"))
        if deferred:
            p.flush_commitments()
            c1, c2 = p.commitment(0), p.commitment(1)
        return p, t.state, c1, c2
    p, st, c1, c2 = build(False)
    pd, std, d1, d2 = build(True)
    assert (c1, c2) == (d1, d2) and c1 != c2 and len(c1) == 32 and st == std
    assert p.get_num_multiplications() == 8 and p.num_committed() == 2
    with pytest.raises(bpg.BpgError) as e:
        p.prove(8, bytes(32))
    assert e.value.status == 7
