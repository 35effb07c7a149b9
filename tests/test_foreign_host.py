"""The drop-in boundary as the reference would use it (INTEGRATION.md sections 1-3): a host that keeps its OWN Merlin transcript, its OWN
constraint system and gadgets - here the pure-Python restatement tests/golden/pyref_r1cs.py, standing in for the Rust code of src/cs_buffer.rs and the
gadget modules - and binds PART 1 of include/bpg.h only: bpg_pedersen_commit for the point part of Prover::commit, bpg_r1cs_prove on the flattened
instance with the 203-byte STROBE state of ITS transcript, bpg_r1cs_verify.  Nothing of PART 2 (the library's own transcript / prover / gadgets) is
involved in producing the proof; the bytes must equal the oracle's on the same flattened instance and the library's own PART-2 assembly of the same
circuit (which a second, independent host must reproduce for the drop-in to be one)."""
import ctypes as C
import numpy as np
import pytest
import bulletproofs_gadgets_amd as bpg
import oracle_lib as O
import assembly_cases as AC
import pyref_r1cs as PR
from gen_assembly_fixtures import PyrefApi

pytestmark = pytest.mark.gpu
L = PR.L


def flatten(p: PR.RecordingProver, with_witness=True):
    """What a host hands to bpg_r1cs_prove: CSR constraint rows with a de-duplicated coefficient table, a_L / a_R / a_O as 32-byte scalars."""
    row_ptr, term_var, term_coef, coef, index = [0], [], [], [], {}
    for lc in p.constraints:
        for v, c in lc.terms:
            c %= L
            if c not in index:
                index[c] = len(coef); coef.append(c)
            term_var.append(v); term_coef.append(index[c])
        row_ptr.append(len(term_var))
    sc = lambda xs: b"".join((x % L).to_bytes(32, "little") for x in xs)
    view = bpg.R1CSInstance()
    view.n, view.q, view.m, view.nnz, view.ncoef = len(p.aL), len(p.constraints), len(p.v), len(term_var), len(coef)
    keep = [np.array(row_ptr, dtype=np.uint64), np.array(term_var, dtype=np.uint32), np.array(term_coef, dtype=np.uint32),
            C.create_string_buffer(sc(coef), max(32 * len(coef), 1))]
    view.row_ptr, view.term_var, view.term_coef = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data
    view.coef = C.cast(keep[3], C.c_void_p).value
    if with_witness:
        for name, xs in (("aL", p.aL), ("aR", p.aR), ("aO", p.aO)):
            buf = C.create_string_buffer(sc(xs), max(32 * len(xs), 1)); keep.append(buf)
            setattr(view, name, C.cast(buf, C.c_void_p).value)
    inst = bpg.FlatInstance(view, v=sc(p.v), v_blinding=sc(p.vb))       # owned copy
    del keep
    return inst


@pytest.mark.parametrize("name", ["cfg2_bounds_check_64", "bounds_check_reference", "mimc_1_block", "mimc_full_last_block", "merkle_2", "merkle_4"])
def test_a_host_with_its_own_merlin_and_assembly_binds_part_1_only(name):
    ctx = bpg.Context(0)
    # ---- the foreign host: own transcript, own commitments, own gadgets (no library object is created for it)
    p, t, coms = AC.build(PyrefApi, name)
    assert p.satisfied()
    values = [(x if x < 2**255 else x % L).to_bytes(32, "little") for x in p.v]
    blinds = [x.to_bytes(32, "little") for x in p.vb]
    # Prover::commit's point part through the C ABI == the host's own Pedersen commitments (src/gadget.rs:31, src/commitments.rs:27,39)
    assert ctx.pedersen_commit(values, blinds) == list(coms)
    inst = flatten(p)
    state = t.state                                                      # 203 bytes of ITS STROBE state after Prover::new and every "V" append
    assert len(state) == 203
    cap = 1
    while cap < inst.n:
        cap *= 2
    ctx.gens_ensure(cap)
    seed = bytes(range(32))
    proof, state_after = ctx.prove_flat(inst, state, inst.v_blinding, seed, 0)           # bpg_r1cs_prove
    # ---- the oracle on the same flattened instance: same bytes, and its verifier accepts on the verifier side (no assignments)
    og = O.Gens(cap)
    oc = O.FlatCircuit(inst.n, inst.m, inst.aL, inst.aR, inst.aO, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)
    rc, want, want_state = O.prove(og, state, oc, inst.v_blinding, seed, O.FLAG_FAST_MSM)
    assert rc == 0 and proof == want and state_after == want_state
    vinst = flatten(p, with_witness=False)
    ocv = O.FlatCircuit(vinst.n, vinst.m, None, None, None, vinst.row_ptr, vinst.term_var, vinst.term_coef, vinst.coef)
    V = b"".join(coms)
    assert O.verify(og, state, ocv, V, proof) == 0
    assert ctx.verify_flat(vinst, state, V, proof) == 0                                  # bpg_r1cs_verify
    bad = bytearray(proof); bad[33] ^= 4
    assert ctx.verify_flat(vinst, state, V, bytes(bad)) in (2, 3)
    # ---- the library's own host mirror (PART 2) assembles the same circuit to the same transcript and proves to the same bytes
    p2, t2, coms2 = AC.build(bpg, name, ctx)
    assert list(coms2) == list(coms) and t2.state == state
    assert p2.prove(bpg.BulletproofGens(ctx, cap), seed) == proof
    ctx.close()
