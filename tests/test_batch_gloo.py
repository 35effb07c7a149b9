"""N > 1 path on CPU: two gloo ranks shard a batch of independent proofs round-robin, prove their share (with the
ORACLE standing in for the GPU - this test is about sharding and the gather, the GPU path is tests/test_gpu_parity.py),
all_gather the proof bytes and every rank verifies the whole batch."""
import os
import subprocess
import sys
import textwrap
import pathlib
import socket


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent('''
    import os, sys, hashlib
    sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import oracle_lib as O, pyref as R
    from bulletproofs_gadgets_amd.batch import shard_indices, gather_proofs
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = R.L
    sc = lambda x: (x %% L).to_bytes(32, "little")
    NUM = 5                                   # odd on purpose: ragged shards
    def instance(i):
        x = int.from_bytes(hashlib.sha256(b"x%%d" %% i).digest(), "little") %% L
        y = int.from_bytes(hashlib.sha256(b"y%%d" %% i).digest(), "little") %% L
        v = [x, y, x * y %% L]
        blind = [int.from_bytes(hashlib.sha256(b"b%%d-%%d" %% (i, k)).digest(), "little") %% L for k in range(3)]
        tv, tc, rows = [], [], [0]
        for kind, idx in ((0, 0), (1, 1), (2, 2)):
            tv += [(kind << 29) | 0, (3 << 29) | idx]; tc += [0, 1]; rows.append(len(tv))
        circ = O.FlatCircuit(1, 3, sc(x), sc(y), sc(x * y), rows, tv, tc, sc(1) + sc(L - 1))
        t = O.Transcript(b"batch-%%d" %% i); t.append(b"dom-sep", b"r1cs v1")
        V = b""
        for a, b in zip(v, blind):
            c = O.pedersen_commit(sc(a), sc(b)); V += c; t.append(b"V", c)
        return circ, t.state, V, b"".join(map(sc, blind))
    gens = O.Gens(4)
    local = {}
    for i in shard_indices(NUM, rank, world):
        circ, st, V, vb = instance(i)
        rc, proof, _ = O.prove(gens, st, circ, vb, bytes([i]) * 32, O.FLAG_FAST_MSM)
        assert rc == 0
        local[i] = proof
    assert sorted(local) == list(range(rank, NUM, world))
    proofs = gather_proofs(local, NUM, O.proof_size(1), dist)
    assert len(proofs) == NUM and all(p is not None for p in proofs)
    for i, p in enumerate(proofs):
        circ, st, V, vb = instance(i)
        assert O.verify(gens, st, circ, V, p) == 0, i
    # a proof shuffled to the wrong slot must not verify
    circ, st, V, vb = instance(0)
    assert O.verify(gens, st, circ, V, proofs[1]) != 0
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank " + str(rank) + " ok\\n"); sys.stdout.flush()      # one write per rank: the two ranks share the pipe
''') % (str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden"))


def test_two_rank_gloo_batch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for attempt in range(2):                  # one retry, with a new port, if the rendezvous itself failed (port race, slow start)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        rendezvous = any(k in r.stderr for k in ("Address already in use", "RendezvousConnectionError", "Connection refused", "timed out", "TCPStore"))
        if r.returncode == 0 or not rendezvous or "AssertionError" in r.stderr:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_shard_indices_cover_every_proof_once():
    from bulletproofs_gadgets_amd.batch import shard_indices
    for world in (1, 2, 3, 8):
        for num in (0, 1, 7, 8, 9, 64):
            seen = sorted(i for r in range(world) for i in shard_indices(num, r, world))
            assert seen == list(range(num))
