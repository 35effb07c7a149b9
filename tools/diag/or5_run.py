import sys, time, shutil, pathlib, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import cli
RES = pathlib.Path("/root/repo/tests/golden/resources")
ctx = bpg.Context(0)
tmp = pathlib.Path(tempfile.mkdtemp())
name = "or5"
for ext in ("gadgets", "inst", "wtns"):
    shutil.copy(RES / ("%s.%s" % (name, ext)), tmp / ("%s.%s" % (name, ext)))
t0 = time.perf_counter()
p, proof = cli.prover(str(tmp / name), ctx=ctx, seed=b"cli-test", rng_seed=bytes(32), quiet=True)
t1 = time.perf_counter()
print("prover", t1 - t0, "s n=", p.get_num_multiplications(), "q=", p.num_constraints(), "m=", p.num_committed(), "proof", len(proof), flush=True)
ok = cli.verifier(str(tmp / name), ctx=ctx, quiet=True)
print("verifier", time.perf_counter() - t1, "s ->", ok, flush=True)
