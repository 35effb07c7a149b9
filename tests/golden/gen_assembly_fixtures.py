#!/usr/bin/env python3
"""Writes tests/golden/assembly.json from the independent Python restatement (pyref_r1cs.py; no product code, no GPU):
for every case of assembly_cases.py and for the reference's README example (tests/golden/resources/example.*, driven through the file
driver's orchestration with every gadget, the constraint system, the commitments and the MiMC hash replaced by the Python ones):
n, q, m, SHA-256 of the canonical constraint rows, of a_L || a_R || a_O, of the committed values and blindings, the commitments, and the
Merlin transcript state after the last commitment.   usage: python tests/golden/gen_assembly_fixtures.py"""
import json
import pathlib
import sys

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(HERE.parent.parent))
import assembly_cases as AC
import pyref_r1cs as PR


class PyrefApi:
    Transcript, Prover = PR.Transcript, PR.RecordingProver
    BoundsCheck, MimcHash256, MerkleTree256 = PR.BoundsCheck, PR.MimcHash256, PR.MerkleTree256
    commit, commit_single, commit_all_single = staticmethod(PR.commit), staticmethod(PR.commit_single), staticmethod(PR.commit_all_single)
    mimc_hash, be_to_scalar = staticmethod(PR.mimc_hash), staticmethod(PR.be_to_scalar)


def record(p, t, coms):
    assert p.satisfied(), "the restated circuit does not satisfy its own constraints"
    d = PR.summary(p)
    d["commitments"] = [c.hex() for c in coms]
    d["transcript_state_sha256"] = __import__("hashlib").sha256(t.t.strobe.state_bytes()).hexdigest()
    return d


def example_through_pyref(stem, seed=b"assembly-example"):
    """cli.prover's orchestration (prover.rs:47-100 restated in bulletproofs_gadgets_amd/cli.py) with the Python constraint system and gadgets."""
    import types
    src = (HERE.parent.parent / "bulletproofs_gadgets_amd" / "cli.py").read_text()
    mod = types.ModuleType("cli_under_pyref")
    header_end = src.index("_VAR = re.compile")
    body = "import hashlib, os, re, sys\n" + src[header_end:]
    conv = lambda x: x if isinstance(x, (bytes, bytearray)) else x.to_bytes(32, "little")      # raw from_bits value: Inequality::compare looks at as_bytes()
    ns = {"BoundsCheck": PR.BoundsCheck, "MimcHash256": PR.MimcHash256, "MerkleTree256": PR.MerkleTree256, "Equality": PR.Equality, "Inequality": PR.Inequality,
          "LessThan": PR.LessThan, "SetMembership": PR.SetMembership, "Prover": PR.RecordingProver, "Transcript": PR.Transcript,
          "commit": PR.commit, "commit_single": PR.commit_single, "mimc_hash": lambda b: conv(PR.mimc_hash(b)), "L": PR.L,
          "be_to_scalar": lambda b: conv(PR.be_to_scalar(b)), "be_to_scalars": lambda b: [conv(x) for x in PR.be_to_scalars(b)],
          "scalar_to_be": lambda s: bytes(reversed(s)), "BulletproofGens": lambda ctx, cap: None, "Context": None,
          "ConstraintBuffer": None, "or_conjunction": None, "Verifier": None}
    mod.__dict__.update(ns)
    exec(compile(body, "cli_under_pyref", "exec"), mod.__dict__)
    import shutil, tempfile
    tmp = pathlib.Path(tempfile.mkdtemp())
    for ext in (".gadgets", ".inst", ".wtns"):
        shutil.copy(str(stem) + ext, tmp / ("example" + ext))
    import os
    cwd = os.getcwd(); os.chdir(tmp)
    try:
        p, _ = mod.prover("example", ctx=object(), seed=seed, rng_seed=bytes(32), quiet=True)
        coms_text = (tmp / "example.coms").read_text()
    finally:
        os.chdir(cwd)
    return p, coms_text


def main():
    out = {}
    for name in AC.CASES:
        p, t, coms = AC.build(PyrefApi, name)
        out[name] = record(p, t, coms)
        print(name, {k: out[name][k] for k in ("n", "q", "m")})
    p, coms_text = example_through_pyref(HERE / "resources" / "example")
    d = record(p, p.transcript, [])
    d["coms_file_sha256"] = __import__("hashlib").sha256(coms_text.encode()).hexdigest()
    d["blinding_seed"] = "assembly-example"
    out["example_gadgets"] = d
    print("example_gadgets", {k: d[k] for k in ("n", "q", "m")})
    (HERE / "assembly.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
