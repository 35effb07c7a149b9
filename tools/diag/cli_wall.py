import subprocess, time, os, shutil, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = tempfile.mkdtemp()
for ext in ("gadgets", "inst", "wtns"):
    shutil.copy(os.path.join(root, "tests/golden/resources/example." + ext), d)
for tool in ("bpg_prover", "bpg_prover", "bpg_prover", "bpg_verifier"):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(root, "bulletproofs_gadgets_amd/bin", tool), "example"], cwd=d, capture_output=True, text=True, env=dict(os.environ, BPG_CLI_TIMING="1"))
    sys.stderr.write(r.stderr)
    print("%s example: %.3f s wall, rc=%d, stdout=%s" % (tool, time.perf_counter() - t0, r.returncode, r.stdout.strip()), flush=True)
