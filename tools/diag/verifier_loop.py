#!/usr/bin/env python3
"""Diagnostic for the one silent SIGSEGV of bpg_verifier seen in round 4: the failing case of tests/test_cli_native.py
(seed 4242, first instance value broken) run `reps` times, one process each, sequentially; prints every exit code that is not 1 with its stderr
(the drivers print a backtrace on a fatal signal).  usage: verifier_loop.py [reps=25]"""
import importlib.util, os, pathlib, subprocess, sys, tempfile
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from bulletproofs_gadgets_amd import build as bpg_build
spec = importlib.util.spec_from_file_location("tcn", str(ROOT / "tests" / "test_cli_native.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
prover_bin, verifier_bin = bpg_build.build_cli()
d = pathlib.Path(tempfile.mkdtemp())
m._random_gadget_files(str(d / "rnd"), 4242)
env = dict(os.environ, BPG_CLI_SEED="fuzz", BPG_CLI_RNG_SEED="11" * 32)
r = subprocess.run([str(prover_bin), "rnd"], cwd=d, env=env, capture_output=True, text=True, timeout=300)
assert r.returncode == 0, r.stderr
lines = (d / "rnd.inst").read_text().splitlines()
name, val = lines[0].split(" = 0x")
lines[0] = "%s = 0x%s" % (name, val[:-1] + ("0" if val[-1] != "0" else "1"))
(d / "rnd.inst").write_text("\n".join(lines) + "\n")
codes = {}
for k in range(reps):
    v = subprocess.run([str(verifier_bin), "rnd"], cwd=d, capture_output=True, text=True, timeout=300)
    codes[v.returncode] = codes.get(v.returncode, 0) + 1
    if v.returncode != 1 or v.stdout.strip() != "false":
        print("run %d: rc %d stdout %r\n%s" % (k, v.returncode, v.stdout, v.stderr[-4000:]))
print("exit codes over %d runs: %s" % (reps, codes))
