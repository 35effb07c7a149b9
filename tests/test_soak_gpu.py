"""Endurance legs inside `pytest -m gpu` (round-4 review, item 3): a short soak of the concurrent proving mix and a loop over the native drivers.

* the soak is tools/diag/soak.py in a process of its own (its six engine contexts, chain pool and serving tables must not share the card's memory
  accounting with this test process): mixed 2^16 / 2^20 proofs on six proving streams, the shared-device kernel variants switching on and off as
  proofs of the other streams come and go, every proof byte-equal to the proof made for its seed alone, host RSS and device memory flat;
* the driver loop runs `bpg_verifier` sixty times back to back, one process each, on a true and on a false statement, and checks exit code and
  stdout every time (reference src/bin/verifier.rs:91-100: the exit code is the contract) - the place where one silent SIGSEGV was seen in round 4
  (DESIGN.md section 9)."""
import os
import pathlib
import re
import subprocess
import sys
import pytest
from bulletproofs_gadgets_amd import build as bpg_build

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_soak_of_the_concurrent_mix_for_twenty_seconds():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "diag" / "soak.py"), "20", "6"], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    m = re.search(r"soak: (\d+) proofs in \d+ s on 6 streams .*mismatches: none; host RSS ([\d.]+) -> ([\d.]+) GB; device memory ([\d.]+) -> ([\d.]+) GB", r.stdout)
    assert m, r.stdout
    proofs, r0, r1, d0, d1 = int(m.group(1)), *map(float, m.groups()[1:])
    assert proofs >= 200                                           # ~55 proofs/s of the 4:1 mix on an MI355X; a stalled stream would show here
    assert r1 <= r0 + 0.5 and d1 <= d0 + 1.0


def _files(tmp_path):
    """a small random gadget file set (the generator of tests/test_cli_native.py), proved once by the native prover"""
    import test_cli_native as tcn
    prover_bin, verifier_bin = bpg_build.build_cli()
    tcn._random_gadget_files(str(tmp_path / "rnd"), 4242)
    env = dict(os.environ, BPG_CLI_SEED="fuzz", BPG_CLI_RNG_SEED="11" * 32)
    r = subprocess.run([str(prover_bin), "rnd"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return verifier_bin


@pytest.mark.gpu
def test_sixty_verifier_processes_in_a_row_keep_their_exit_codes(tmp_path):
    verifier_bin = _files(tmp_path)
    good = (tmp_path / "rnd.inst").read_text()
    lines = good.splitlines()
    name, val = lines[0].split(" = 0x")
    lines[0] = "%s = 0x%s" % (name, val[:-1] + ("0" if val[-1] != "0" else "1"))
    bad = "\n".join(lines) + "\n"
    seen = {}
    for k in range(60):
        want = (0, "true") if k % 2 == 0 else (1, "false")
        (tmp_path / "rnd.inst").write_text(good if k % 2 == 0 else bad)
        v = subprocess.run([str(verifier_bin), "rnd"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        got = (v.returncode, v.stdout.strip())
        seen[got] = seen.get(got, 0) + 1
        assert got == want, "run %d: %r instead of %r (all so far: %r)\n%s" % (k, got, want, seen, v.stderr[-4000:])
    assert seen == {(0, "true"): 30, (1, "false"): 30}
