#!/usr/bin/env python3
"""bench.py - R1CS constraints/sec of the prove step on BASELINE.json's 2^20 MiMC-Merkle circuit.

One "step" = one Prover::prove (reference src/bin/prover.rs:93) of the reference's own 2^20 circuit - the full
512-leaf MiMC Merkle tree of src/merkle_tree/merkle_tree_gadget.rs:473-545 (n = 993,384 multipliers padded to
N = 2^20, q = 1,986,769 constraints, m = 512 commitments) - with the flattened instance and the generator tables
already resident in HBM.  Steps are independent proofs (own seed), dealt round-robin to --streams proving streams per GPU (engine
contexts: own HIP stream and proving thread, generator tables shared); their serial TranscriptRng chains (0.3 s of host work each,
upstream-exact) are drawn by --chain-workers host threads, shared out over the streams (defaults: 14 streams with one chain thread
each; `--streams 1 --chain-workers 1` = one stream, one host thread, reported as `single_stream`); every chain of the K timed
steps starts and ends inside the timed region.

`--gpus N` (N > 1) without a torchrun environment: this process spawns N ranks itself (fresh child processes, before anything
touches torch or the GPU) and relays rank 0's line.  Under torchrun it is one rank: one process per GPU, each rank proves its
own independent proof per step (weak scaling); the only collective is an RCCL all_gather of the finished proof bytes.
A second, strong-scaling leg proves a fixed batch of 8 independent proofs sharded round-robin over the ranks.

Prints ONE JSON line on rank 0 (see the driver contract); diagnostics go to stderr.
"""
import argparse
import hashlib
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

METRIC = "R1CS constraints/sec (prove), 2^20-constraint MiMC-Merkle, 1/2/4/8 GPU"
SWEEPS = ("k_bucket_chunks",)                              # the bucket sweep (csrc/hip/k_msm.cuh)
ENGINE = {"profile": "serving", "blocking_sync": True}       # bpg_config of every engine context (include/bpg.h bpg_ctx_create_ex)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--leaves", type=int, default=512, help="leaves of the full MiMC Merkle tree (512 = the reference's 2^20 circuit)")
    ap.add_argument("--leaf-seed", type=int, default=None, help="leaves of the tree: unset = the reference's own instance (512 times the SAME leaf, merkle_tree_gadget.rs:476-520: every "
                    "node of a level then carries the same values, and the equal-scalar merging of A_I / A_O collapses 2.98 M terms into ~35 k); an integer = distinct seeded leaves "
                    "(only the wiring of a MiMC round repeats a value: 43 %% of A_I's terms merge, none of A_O's)")
    ap.add_argument("--baseline-leaves", type=int, default=512, help="leaves of the CPU-baseline tree (512 = the headline circuit itself, about 150 s on one core; "
                    "64 -> N = 2^17, about 17 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-workers", type=int, default=14, help="chain threads per rank (bpg_ctx_set_chain_workers, shared out over the proving streams) = how many "
                    "TranscriptRng chains are drawn side by side; sized for the ~16 cores a GPU has to itself on an 8-GPU host; 1 = one host thread, the chain of "
                    "step i+1 under the kernels of step i")
    ap.add_argument("--streams", type=int, default=20, help="proving streams per GPU for the headline (at most one per step): that many engine contexts (own HIP stream, "
                    "proving thread, share of the chain workers; generator tables shared) take the steps round-robin. With one chain thread per stream every proof "
                    "runs A_I, A_O and most of S under its own chain, so the GPU works through the 0.3 s the first chains take")
    ap.add_argument("--chain-lanes", type=int, default=1, help="streams each chain thread draws in lockstep (bpg_ctx_set_chain_lanes, 1..8: the sponges of several proofs in the "
                    "lanes of ZMM registers): with 8 a chain takes 27 %% longer and a core draws six times as many; the headline keeps 1 (shortest chains)")
    ap.add_argument("--chain-pool", default="auto", help="draw the chains of ALL proving streams on one shared pool (bpg_chain_pool_create) instead of a chain worker per stream: "
                    "comma-separated THREADSxLANES groups, e.g. 14x1,1x6 = fourteen threads that draw one chain each (0.30 s at 2^20) and one thread that draws six in "
                    "lockstep (0.38 s): twenty chains at once on fifteen cores; `auto` (default) sizes it for the cores this rank has (plan_chain_pool), \"\" = a chain "
                    "worker per stream (--chain-workers / --chain-lanes)")
    ap.add_argument("--no-prefetch", action="store_true", help="draw every chain inside its own prove call (round-1 behaviour): the GPU idles while the host draws")
    ap.add_argument("--batch", type=int, default=8, help="strong-scaling leg: this many independent proofs in total, sharded round-robin over the ranks (0 = skip)")
    ap.add_argument("--kernel-profile", action="store_true", help="list every kernel's HIP-event total of one untimed proof in the line")
    ap.add_argument("--headline-only", action="store_true", help="only the warmup and timed steps (no verify / expanded-blinding / in-flight / batch / CPU legs): "
                    "the process then launches nothing but the headline's kernels, which is what the rocprofv3 passes of tools/profile_round.sh want")
    ap.add_argument("--in-flight", type=int, default=6, help="throughput leg: proving streams (engine contexts, one proving thread each) on ONE GPU (0 = skip)")
    ap.add_argument("--in-flight-workers", type=int, default=1, help="throughput leg: chain threads per context")
    ap.add_argument("--in-flight-lanes", type=int, default=8, help="throughput leg: streams per chain thread in lockstep (1..8)")
    ap.add_argument("--in-flight-steps", type=int, default=24, help="throughput leg: proofs per context")
    ap.add_argument("--in-flight-only", action="store_true", help="only the throughput leg (for profiling the concurrent kernel mix)")
    ap.add_argument("--profile", choices=("serving", "oneshot"), default="serving", help="bpg_config.profile of every engine context: serving (the headline: a long-lived "
                    "prover, 51.5 GB of fold tables per device) or oneshot (what a bare bpg_ctx_create gives a drop-in caller: 3 GB of tables)")
    ap.add_argument("--plan-only", action="store_true", help="print what --gpus N would do on this node - per rank: device, NUMA node, CPU set, chain-pool lanes, expected HBM and "
                    "pinned host memory - from the planner the real run uses, WITHOUT touching a GPU; a node with fewer cards is planned on an assumed 2-socket topology")
    ap.add_argument("--lone-proof", action="store_true", help="with --headline-only: also prove once ALONE on the device after the timed steps and report `lone_proof` "
                    "(phase times of the library clock, sum of the kernels' HIP-event durations, launches) - what the one-shot child of the default run adds to `value_one_shot`")
    ap.add_argument("--no-one-shot-leg", action="store_true", help="skip `value_one_shot` (the timed steps once more under the one-shot profile, in a child process)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher (N > 1 without torchrun)
def spawn_ranks(args):
    """Start `--gpus N` ranks as fresh child processes (python -m torch.distributed.run, rendezvous on 127.0.0.1) BEFORE this process has
    imported torch or touched the GPU; the children inherit stdout/stderr, so rank 0's JSON line is this command's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py")] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("bench.py: spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ host placement
def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def pin_near_gpu(torch, device_index):
    """Restrict this rank (and the threads it creates later: chain workers, proving threads) to the CPUs of the NUMA node its GPU hangs off.
    Returns a description for the JSON line; any failure leaves the affinity alone."""
    info = {"numa_node": None, "cpus_allowed": len(os.sched_getaffinity(0))}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        info["pci"] = bdf
        node = int(pathlib.Path("/sys/bus/pci/devices/%s/numa_node" % bdf).read_text())
        info["numa_node"] = node
        if node < 0:
            return info
        cpus = _cpulist(pathlib.Path("/sys/devices/system/node/node%d/cpulist" % node).read_text()) & os.sched_getaffinity(0)
        cores = set()
        for c in cpus:
            sib = _cpulist(pathlib.Path("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read_text())
            if c == min(sib & cpus):
                cores.add(c)
        if len(cpus) >= 2:
            # every hardware thread of the node: the scheduler fills idle cores before it doubles up on one, and a rank's ~15 busy threads
            # (chain workers, proving threads) plus the runtime's helpers must never be squeezed onto fewer cores than that
            os.sched_setaffinity(0, cpus)
            info["cpus_allowed"] = len(cpus)
            info["cores_allowed"] = len(cores)
            info["pinned"] = "the CPUs of NUMA node %d" % node
    except Exception as e:      # noqa: BLE001 - placement is an optimisation, never a reason to fail
        info["error"] = repr(e)
    return info


def current_cpu():
    try:
        import ctypes
        return int(ctypes.CDLL(None).sched_getcpu())
    except Exception:       # noqa: BLE001
        return None


def cpu_quota():
    """CPUs this process may use at once according to its cgroup (cpu.max of cgroup v2, cfs quota of v1); None = no quota.  A box that shows every
    logical CPU of the host but is throttled to a share of them (the one-GPU boxes of the pool: 16) must be planned for the share."""
    try:
        txt = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt and txt[0] != "max":
            return max(1, int(int(txt[0]) / int(txt[1])))
        return None
    except Exception:       # noqa: BLE001
        pass
    try:
        q = int(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = int(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return max(1, q // per) if q > 0 else None
    except Exception:       # noqa: BLE001
        return None


def cores_for_rank(cores_on_node, ranks_on_node, quota=None):
    """Cores one rank can keep busy: the cores of its NUMA node shared with the other ranks placed there, capped by the cgroup quota (which all
    ranks of the process group share)."""
    per = max(1, cores_on_node // max(1, ranks_on_node))
    if quota is not None:
        per = min(per, max(1, quota // max(1, ranks_on_node)))
    return per


def plan_chain_pool(cores_per_rank, n_streams, reserve=2):
    """Lane counts of the chain-pool threads of one rank: every proving stream's first chain must be drawn AT ONCE (a step cannot finish before its
    chain has), a chain thread needs a core to itself, `reserve` cores stay free for the proving threads and the runtime.  One lane per thread where
    the cores allow (0.30 s per chain at 2^20); beyond that, as few lockstep threads as possible take the rest (0.38 s per chain, up to 8 lanes)."""
    T = max(1, cores_per_rank - reserve)
    if T >= n_streams:
        return [1] * n_streams
    k = -(-(n_streams - T) // 7)                              # lockstep threads: each replaces one single-lane thread and adds up to 7 chains
    k = min(k, T)
    singles = T - k
    rest = n_streams - singles
    lanes = [1] * singles
    for j in range(k):
        share = min(8, -(-(rest - sum(lanes[singles:])) // (k - j)))
        lanes.append(max(1, share))
    return lanes                                              # sum(lanes) < n_streams only when 8 * T < n_streams: the remaining chains queue


# ------------------------------------------------------------------------------------------------ memory model and the plan of a node (--plan-only)
def memory_model(n, N, q, m, profile, streams, nnz=None, merged_terms=0):
    """Bytes ONE rank holds in HBM and in pinned host memory at the headline arrangement: an analytic mirror of the allocations of csrc/engine.hip (the buffer
    names are the engine's), checked against what the card reports in the bench line (`hbm_in_use` beside `hbm_model`).  n multipliers padded to N, q
    constraints, m commitments; `streams` proving streams (engine contexts) with a resident copy of the circuit and two blinding slabs each."""
    S, P_NIELS, P_EXT, P_PN = 32, 96, 128, 128
    nnz = nnz if nnz is not None else 3 * q                     # MiMC circuits: about three terms per constraint
    W, nb = 16, 1 << 15                                         # shared-device windows: 16 bits (17 x 2^14 for a proof alone: the same bytes within 6 %)
    al = lambda x: (x + 255) & ~255

    def msm_arena(total, live):
        mub = live * W
        slots = -(-mub // 34) * 2 * P_EXT                        # two partial sums per chunk of the sweep; 34-entry chunks (a proof alone), 64 while the device is shared
        return max(al(total * W * 2) + al(live * W * 4), al(slots)) + al(live * W * 4)
    # device-wide, shared by the contexts of the process: [G | H] and the odd multiples of the first fold
    gens = 2 * N * P_NIELS
    tables = (255 if profile == "serving" else 15) * gens
    # per proving stream
    calls = [(3 * n + 2, 3 * n + 2 - merged_terms), (2 * n + 1, 2 * n + 1), (2 * N + 2, 2 * N + 2)]      # A_I + A_O, the largest piece of S, a round of the first fold group
    arena = max([msm_arena(t, l) for t, l in calls] + [al((2 * 4096 + 1) * 64 * P_EXT) + (2 * 4096 + 1) * 64 * 8 * P_PN, (3 * n + m + 1 + q + 2 + N) * S])
    buckets = 3 * W * nb * P_EXT                                # up to three sums in one launch (A_I, A_O, S when the chain was drawn ahead)
    partial = 2 * 3 * W * (nb // 8) * P_EXT
    tiles = -(-2 * n // 4096)
    sortws = 3 * (3 * W * 256 * tiles + 1) * 4 + 3 * W * nb * 4
    prove = 2 * n * S + 3 * N * S + 4 * (N // 2) * S + 3 * (N // 2) * P_NIELS + 2 * (N >> 3) * P_EXT      # s_L|s_R, y^-i, l, r, expanded scalars, folded tables A|B, fold scratch
    slabs = 2 * (2 * n * 64)
    circuit = 3 * n * S + (3 * n + m + 2) * 8 + 2 * nnz * 4
    per_stream = arena + buckets + partial + sortws + prove + slabs + circuit
    runtime = 1.2e9                                             # HIP runtime, code objects, torch: measured on an idle context
    hbm = gens + tables + streams * per_stream + runtime
    pinned = streams * (2 * (2 * n * 64) + (16 << 20) + (1 << 20))      # two blinding slabs, the upload bounce slots, window-sum slots
    return {"hbm_GB": hbm / 1e9, "pinned_host_GB": pinned / 1e9, "per_stream_GB": per_stream / 1e9, "tables_GB": (gens + tables) / 1e9,
            "arena_GB": arena / 1e9, "streams": streams, "profile": profile}


def discover_topology(gpus):
    """The node as sysfs shows it, WITHOUT touching a GPU: AMD display / accelerator functions with their NUMA node, the cores of every node, host RAM,
    the cgroup CPU quota.  None when fewer than `gpus` devices are visible (the build container: none)."""
    devs = []
    try:
        for d in sorted(pathlib.Path("/sys/bus/pci/devices").iterdir()):
            try:
                if (d / "vendor").read_text().strip() != "0x1002":
                    continue
                cls = (d / "class").read_text().strip()
                if not (cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200")):
                    continue
                node = int((d / "numa_node").read_text())
                # a container may be given some of the node's cards only: a card counts when its render node can be opened (what the runtime will find)
                render = sorted((d / "drm").glob("renderD*")) if (d / "drm").is_dir() else []
                usable = None if not render else any(os.access("/dev/dri/" + r.name, os.R_OK | os.W_OK) for r in render)
                devs.append({"pci": d.name, "numa_node": max(node, 0), "usable": usable})
            except Exception:       # noqa: BLE001
                continue
    except Exception:       # noqa: BLE001
        return None
    if any(x["usable"] for x in devs):
        devs = [x for x in devs if x["usable"]]
    for x in devs:
        x.pop("usable", None)
    if len(devs) < gpus:
        return None
    nodes = {}
    for nd in sorted(pathlib.Path("/sys/devices/system/node").glob("node[0-9]*")):
        cpus = _cpulist((nd / "cpulist").read_text())
        cores = set()
        for c in cpus:
            try:
                sib = _cpulist(pathlib.Path("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read_text())
            except Exception:       # noqa: BLE001
                sib = {c}
            if c == min(sib & cpus):
                cores.add(c)
        nodes[int(nd.name[4:])] = {"cpus": len(cpus), "cores": len(cores)}
    ram = None
    try:
        for line in pathlib.Path("/proc/meminfo").read_text().splitlines():
            if line.startswith("MemTotal:"):
                ram = int(line.split()[1]) * 1024 / 1e9
    except Exception:       # noqa: BLE001
        pass
    return {"source": "sysfs", "devices": devs[:gpus], "nodes": nodes, "host_ram_GB": ram, "cpu_quota": cpu_quota(), "hbm_GB": 288.0}


def synthetic_topology(gpus, numa_nodes=2, cores_per_node=64, host_ram_GB=1536.0, hbm_GB=288.0, cpu_quota=None):
    """An MI355X node as BASELINE config 5 assumes it when no such node is at hand: `gpus` cards spread evenly over `numa_nodes` sockets."""
    return {"source": "assumed", "devices": [{"pci": None, "numa_node": r * numa_nodes // gpus} for r in range(gpus)],
            "nodes": {k: {"cpus": 2 * cores_per_node, "cores": cores_per_node} for k in range(numa_nodes)}, "host_ram_GB": host_ram_GB, "cpu_quota": cpu_quota, "hbm_GB": hbm_GB}


def plan_node(gpus, streams, steps, profile, topology, n=993384, N=1 << 20, q=1986769, m=512):
    """What `bench.py --gpus N` will do on this node, rank by rank, from the planner the real run uses (cores_for_rank, plan_chain_pool) - no GPU call."""
    ranks = []
    n_streams = max(1, min(streams, steps))
    for r in range(gpus):
        node = topology["devices"][r]["numa_node"]
        sharing = sum(1 for d in topology["devices"][:gpus] if d["numa_node"] == node)
        quota = topology.get("cpu_quota")
        per_rank = cores_for_rank(topology["nodes"][node]["cores"], sharing, None if quota is None else quota * sharing // max(1, gpus))
        lanes = plan_chain_pool(per_rank, n_streams)
        mem = memory_model(n, N, q, m, profile, n_streams)
        ranks.append({"rank": r, "device": r, "pci": topology["devices"][r]["pci"], "numa_node": node, "ranks_on_node": sharing, "cores_for_rank": per_rank,
                      "cpu_set": "the CPUs of NUMA node %d" % node, "chain_pool_lanes": lanes, "chain_threads": len(lanes), "chains_at_once": sum(lanes),
                      "proving_streams": n_streams, "hbm_expected_GB": round(mem["hbm_GB"], 1), "pinned_host_expected_GB": round(mem["pinned_host_GB"], 2)})
    pinned = sum(x["pinned_host_expected_GB"] for x in ranks)
    fits = {"hbm_per_card": all(x["hbm_expected_GB"] <= topology["hbm_GB"] for x in ranks),
            "pinned_host": topology.get("host_ram_GB") is None or pinned <= 0.5 * topology["host_ram_GB"],
            "a_core_per_chain_thread": all(x["chain_threads"] + 2 <= x["cores_for_rank"] or x["cores_for_rank"] <= 2 for x in ranks)}
    return {"plan_only": True, "gpus": gpus, "profile": profile, "steps": steps, "topology": {k: v for k, v in topology.items() if k != "devices"},
            "ranks": ranks, "pinned_host_total_GB": round(pinned, 1), "fits": fits,
            "note": "no GPU was touched; HBM and pinned memory from bench.memory_model (csrc/engine.hip's allocations restated), checked against the card in the real line "
                    "(hbm_in_use beside hbm_model)"}


def host_description():
    model = None
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name:"):
                model = line.split(":", 1)[1].strip()
    except Exception:       # noqa: BLE001
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count()}


def source_hash():
    """Hash of the product sources the kernels are built from: a stored PMC profile is only quoted when it was taken from these sources."""
    h = hashlib.sha256()
    for p in sorted((ROOT / "bulletproofs_gadgets_amd" / "csrc").rglob("*")):
        if p.is_file() and p.suffix in (".hip", ".cuh", ".hpp", ".inc"):          # what libbpg_hip.so is built from (cli_main.cpp is the file driver over the C ABI)
            h.update(p.name.encode()); h.update(p.read_bytes())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ secondary legs
def cpu_baseline(ctx, bpg, workloads, leaves, headline):
    """Time the single-threaded CPU oracle (upstream's algorithms: constant-time Straus for A_I/A_O/S/T, Straus/Pippenger for
    L/R, two-point folds), built with -march=native on this host, on a full MiMC Merkle tree - by default the headline circuit itself."""
    import oracle_lib as O
    lib_used = O.use_native()
    if headline is not None and leaves == headline["leaves"]:
        a, inst, res = headline["a"], headline["inst"], headline["res"]
        own = False
    else:
        a = workloads.merkle_full_tree(ctx, leaves=leaves, seed=7)
        inst = a.prover.instance()
        ctx.gens_ensure(a.gens_capacity)
        res = ctx.upload(inst)
        own = True
    oc = O.FlatCircuit(inst.n, inst.m, inst.aL, inst.aR, inst.aO, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)
    t0 = time.perf_counter()
    g = O.Gens(a.gens_capacity)
    t_gens = time.perf_counter() - t0
    seed = bytes(range(32))
    t0 = time.perf_counter()
    rc, proof, _ = O.prove(g, a.transcript.state, oc, inst.v_blinding, seed, 0)
    dt = time.perf_counter() - t0
    assert rc == 0
    # the same instance through the GPU must give the same bytes (keeps the baseline honest about what it computes)
    gp, _ = res.prove(a.transcript.state, inst.v_blinding, seed, 0)
    if own:
        res.free()
    assert gp == proof, "cpu_baseline: GPU and oracle proofs differ"
    return {"value": inst.q / dt, "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": "oracle (single thread, upstream algorithms, -O3 -march=native on this host) on the full %d-leaf MiMC Merkle tree: n=%d, N=%d, q=%d, "
                      "%.1f s prove (generators %.1f s, not counted); the same instance proved on the GPU byte-identically"
                      % (leaves, inst.n, a.gens_capacity, inst.q, dt, t_gens),
            "seconds": dt, "library": os.path.basename(lib_used), "full_size": leaves == 512}


def in_flight_throughput(bpg, ctx0, res0, inst, state, capacity, device, n_ctx, steps, workers=1, lanes=1, restore_workers=1):
    """Throughput figure (never `value`): B independent proofs in flight on one GPU, one engine context + host thread each.
    The serial TranscriptRng chain of one proof then overlaps the kernels of the others."""
    import threading
    ctxs, ress = [ctx0], [res0]
    for _ in range(n_ctx - 1):
        c = bpg.Context(device, **ENGINE)
        c.gens_ensure(capacity)
        ctxs.append(c); ress.append(c.upload(inst))
    for c in ctxs:
        c.set_chain_workers(workers); c.set_chain_lanes(lanes)
    ahead = workers * lanes                                  # chains queued beyond the proof being proved (workers * lanes + 1 streams may be alive)
    start = threading.Barrier(n_ctx + 1)
    errs, done_at = [], []

    def worker(k):
        try:
            seeds = [bytes([k, i]) + bytes(30) for i in range(steps)]
            ress[k].prove(state, inst.v_blinding, bytes([k, 255]) + bytes(30), 0)       # warm-up (workspace allocation)
            start.wait()
            # each context sequences its proofs as the headline does: the chain of its next proof is queued on its chain worker before the
            # current one is proved; under load a proof waits for the GPU longer than a chain takes, so the chain is complete when its
            # proof starts and A_I, A_O, S go through one multiscalar pass instead of four
            queued = 0
            for i in range(steps):
                while queued < steps and queued <= i + ahead:
                    ctxs[k].blinding_begin(state, inst.v_blinding, seeds[queued], inst.n)
                    queued += 1
                ress[k].prove(state, inst.v_blinding, seeds[i], 0)
                done_at.append(time.perf_counter())
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
            try:
                start.abort()
            except Exception:
                pass
    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for r in ress[1:]:
        r.free()
    for c in ctxs[1:]:
        c.close()
    ctx0.set_chain_lanes(1); ctx0.set_chain_workers(max(1, restore_workers))
    if errs:
        raise RuntimeError(errs[0])
    # all contexts start their first chains together, so nothing completes for the first 0.3 - 0.4 s and the first proofs of every context then
    # complete in a bunch: the sustained rate is taken from the completion that follows that bunch (one proof per context) to the last one
    done_at.sort()
    skip = min(n_ctx, len(done_at) - 2) if len(done_at) > 2 else 0
    sustained = (len(done_at) - 1 - skip) / (done_at[-1] - done_at[skip]) if len(done_at) > skip + 1 and done_at[-1] > done_at[skip] else n_ctx * steps / dt
    return {"proofs_in_flight": n_ctx, "proofs": n_ctx * steps, "seconds": dt, "value": inst.q * sustained, "ms_per_proof": 1e3 / sustained,
            "unit": "constraints/s", "whole_run": {"value": inst.q * n_ctx * steps / dt, "ms_per_proof": dt / (n_ctx * steps) * 1e3},
            "host_threads": {"chain": n_ctx * workers, "lanes_per_chain_thread": lanes, "proving": n_ctx},
            "rate": "proofs completed after the first %d completions / time from that completion to the last" % skip}


def throughput_variants(bpg, workloads, device, inst0, state0, capacity, n_ctx, steps, workers, lanes, leaves):
    """The throughput leg once more under the settings that say where the equal-scalar merging of A_I / A_O (csrc/hip/k_merge.cuh) gets its gain:
      merge_per_proof           BPG_MERGE=2: the grouping is redone inside EVERY proof (a host that proves each witness once pays that)
      no_merging                BPG_MERGE=0: term by term, as round 4 proved
      distinct_leaves(_no_merging)  the same tree over 512 DIFFERENT seeded leaves: only the wiring of a MiMC round repeats a value (43 % of A_I's terms merge)
    The reference's own instance (`value`, `throughput`) hashes 512 EQUAL leaves (merkle_tree_gadget.rs:476-520): its 2.98 M terms of A_I and A_O carry ~35 k
    distinct scalars.  Same proof bytes under every setting (tests/test_gpu_parity.py::test_equal_scalar_merging_changes_no_byte)."""
    out = {}
    distinct = None
    for name, env, seeded in (("merge_per_proof", "2", False), ("no_merging", "0", False), ("distinct_leaves", None, True), ("distinct_leaves_no_merging", "0", True)):
        old = os.environ.get("BPG_MERGE")
        try:
            if env is not None:
                os.environ["BPG_MERGE"] = env
            c = bpg.Context(device, **ENGINE)
            try:
                if seeded:
                    if distinct is None:
                        a = workloads.merkle_full_tree(c, leaves=leaves, seed=7)
                        distinct = (a.prover.instance(), a.transcript.state)
                    inst, state = distinct
                else:
                    inst, state = inst0, state0
                c.gens_ensure(capacity)
                res = c.upload(inst)
                res.prove(state, inst.v_blinding, bytes(32), 0)
                sch = c.schedule()
                r = in_flight_throughput(bpg, c, res, inst, state, capacity, device, n_ctx, steps, workers=workers, lanes=lanes)
                out[name] = {"ms_per_proof": r["ms_per_proof"], "value": r["value"], "proofs": r["proofs"], "merged_groups": sch["merged_last"],
                             "terms_merged_away": sch["merged_skipped_last"], "merge_ms": sch["merge_ms_last"]}
                res.free()
            finally:
                c.close()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)}
        finally:
            if old is None:
                os.environ.pop("BPG_MERGE", None)
            else:
                os.environ["BPG_MERGE"] = old
    out["note"] = ("sustained ms per proof of the throughput leg (%d streams x %d proofs) under BPG_MERGE=2 / 0 and on a tree of distinct leaves; merge_ms = wall time of one "
                   "grouping of A_I and A_O (hash table, scans, one point addition per merged-away term, normalisation)" % (n_ctx, steps))
    return out


def file_config(ctx, workloads, which):
    """A configuration that exists as FILES, assembled by the file driver (cli.prover(assemble_only): parsing, commitments and gadget assembly as
    `prover NAME` does them, reference src/bin/prover.rs:47-91) and handed over just before Prover::prove:
      example  BASELINE.json config 1: the reference's own example.gadgets / .inst / .wtns (reference example.gadgets:1-9; the repo keeps its three
               input files as test data under tests/golden/resources/): n = 14,988, q = 30,007, m = 33
      path20   SURVEY.md section 8(d) cfg 4b: one depth-20 Merkle authentication path (workloads.merkle_path_files): n = 39,852, N = 2^16"""
    import shutil
    import tempfile
    from bulletproofs_gadgets_amd import cli
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)                                                       # the transcript label is the NAME argument (prover.rs:49-52): a fixed relative name
        try:
            if which == "example":
                for ext in ("gadgets", "inst", "wtns"):
                    shutil.copy(ROOT / "tests" / "golden" / "resources" / ("example." + ext), "example." + ext)
            else:
                workloads.merkle_path_files("path20", depth=20)
            p, t = cli.prover(which, ctx=ctx, seed=b"bench", rng_seed=bytes(32), quiet=True, two_pass=False, assemble_only=True)
        finally:
            os.chdir(cwd)
    n = p.get_num_multiplications()
    cap = 1
    while cap < n:
        cap *= 2
    return workloads.Assembled(p, t, [p.commitment(i) for i in range(p.num_committed())], cap, None)


def small_configs(bpg, workloads, device):
    """Secondary: BASELINE.json's small configurations (parity cases, never `value`) - median prove / verify time of a lone proof under the library's
    default one-shot profile and under the serving profile of this benchmark (8-bit window tables for circuits up to 2^14 multipliers)."""
    out = {}
    for prof, kw in (("one_shot", {}), ("serving", dict(ENGINE))):
        c = bpg.Context(device, **kw)
        rows = {}
        try:
            for name, mk in (("cfg2_bounds_check_64", lambda: workloads.bounds_check_64(c, seed=0)),
                             ("cfg1_example_gadgets", lambda: file_config(c, workloads, "example")),
                             ("cfg1_size_merkle8_2^14", lambda: workloads.merkle_full_tree(c, leaves=8, seed=7)),
                             ("cfg3_mimc_preimage_2^16", lambda: workloads.mimc_preimage(c, nbytes=2130, seed=0, label=b"MiMCHash")),
                             ("cfg4b_merkle_path_depth20", lambda: file_config(c, workloads, "path20"))):
                a = mk(); inst = a.prover.instance(); state = a.transcript.state
                c.gens_ensure(a.gens_capacity); res = c.upload(inst)
                for i in range(4):
                    res.prove(state, inst.v_blinding, bytes(32), 0)
                ts = []
                for i in range(7):
                    t0 = time.perf_counter(); proof, _ = res.prove(state, inst.v_blinding, bytes([i + 1]) * 32, 0); ts.append(time.perf_counter() - t0)
                tm = res.prove(state, inst.v_blinding, bytes(32), 0, timings=True)[2]
                coms = b"".join(a.commitments)
                t0 = time.perf_counter()
                oks = [res.verify(state, coms, proof) for _ in range(5)]
                dv = (time.perf_counter() - t0) / 5
                dt = sorted(ts)[len(ts) // 2]
                rows[name] = {"n": inst.n, "N": a.gens_capacity, "q": inst.q, "prove_ms": dt * 1e3, "constraints_per_s": inst.q / dt, "host_chain_ms": tm["rng_host"],
                              "ipa_ms": tm["ipa"], "verify_ms": dv * 1e3, "verified": all(o == 0 for o in oks)}
                res.free()
        finally:
            c.close()
        out[prof] = rows
    out["note"] = "one proof at a time, chain drawn inside the call; median of 7; not the headline"
    return out


def end_to_end(bpg, workloads, ctx, capacity, expect, seed):
    """Untimed, secondary: one proof from scratch - 512 commitments, host assembly of the reference's 512-leaf tree, flatten, upload, prove -
    as upstream orders it, and with the TranscriptRng chain started right after the commitments (bpg_prover_start_blinding, include/bpg.h)."""
    leaves = 512
    leaf_be = [bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")] * leaves
    root = bpg.be_to_scalar(bytes.fromhex("038c137beec8e2edfb5c48cbd063f04e569139d2221a4eb7befb85aa1bf8ba40"))
    pattern = workloads.full_tree_pattern(leaves)
    res = {}
    for key, early in (("sequential_s", False), ("chain_beside_assembly_s", True)):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            t = bpg.Transcript(b"MerkleTree")
            p = bpg.Prover(ctx, t)
            _, _, wvars = workloads.commit_all_single(p, leaf_be, [workloads.blinding("cfg4-None", i) for i in range(leaves)])
            if early:
                p.start_blinding(seed, capacity)
            bpg.MerkleTree256(root, [], workloads.vars_to_lc(wvars), pattern).prove(p, [], [])
            proof = p.prove(bpg.BulletproofGens(ctx, capacity), seed)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            if expect is not None and proof != expect:
                raise RuntimeError("end-to-end proof differs from the resident-circuit proof of the same seed")
        res[key] = best
    res["note"] = ("commit + assemble + flatten + upload + prove of one 2^20 proof (best of 2), generators resident; same proof bytes both ways and as the timed "
                   "step of the same seed; not the headline")
    return res


# ------------------------------------------------------------------------------------------------ one rank
def one_shot_leg(args):
    """`value_one_shot`: the SAME K timed steps (same command, streams, chain pool, seeds) with every context under the ONE-SHOT profile - what a
    drop-in caller of bpg_ctx_create(device) gets: first fold on width-5 NAF tables of scalars cut in two (3.0 GB at 2^20), no 8-bit tail tables.  Run as
    a child process: tables are per process and device, and the serving tables this process holds would count against the one-shot budget."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--headline-only", "--lone-proof", "--profile", "oneshot", "--no-one-shot-leg", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--leaves", str(args.leaves), "--streams", str(args.streams), "--chain-workers", str(args.chain_workers), "--chain-lanes", str(args.chain_lanes),
           "--chain-pool", args.chain_pool]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BPG_PROFILE"):
        env.pop(k, None)
    parent_gb = None
    try:
        import torch
        free_b, total_b = torch.cuda.mem_get_info(0)
        parent_gb = (total_b - free_b) / 1e9                    # what THIS process still holds on the card (serving tables, one stream's workspace)
    except Exception:       # noqa: BLE001
        pass
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "one-shot leg failed (rc %d): %s" % (r.returncode, r.stderr[-600:])}
    d = json.loads(lines[-1])
    hbm = d.get("hbm_in_use")
    if hbm and parent_gb is not None:
        hbm = dict(hbm, in_use_GB=hbm["in_use_GB"] - parent_gb, device_in_use_GB=hbm["in_use_GB"], parent_process_GB=parent_gb,
                   note="device memory the one-shot child process holds with all its proving streams alive = what the card reports in use minus what the "
                        "parent (this serving run, idle meanwhile) held when it started the child")
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "hbm_in_use": hbm,
            "completions": d.get("completions"), "schedule": d.get("schedule"), "busy_cores_avg": d["config"]["host_threads_per_gpu"]["busy_cores_avg"],
            "lone_proof": d.get("lone_proof"),
            "note": "the same timed steps with every engine context under the one-shot profile (bpg_ctx_create's default; blocking stream waits as in the headline), "
                    "in a child process of this run while this process idles; `value` of the line is the serving profile"}


def run_rank(args):
    ENGINE["profile"] = args.profile
    # the throughput leg runs a dozen engine streams: 8 hardware queues instead of the runtime's default 4 measured 8 % more proofs/s
    # (measured in round 2); read by the HIP runtime at initialisation, so set before torch / the library load it
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # every context of this benchmark is created with bpg_ctx_create_ex (ENGINE below): the SERVING profile (a long-lived prover: 51.5 GB of fold
    # tables per device, built once) and blocking stream waits - proving threads sleep instead of spinning; with a dozen of them beside a dozen
    # chain threads the spinning costs the chains their cores (measured: 51 -> 54.5 M constraints/s at 14 streams)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py spawn them: run it without torchrun)" % (args.gpus, world))
    # BPG_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (ranks then share devices and the proof bytes
    # travel over gloo on CPU tensors); the driver's runs use the default: one rank per GPU, RCCL ("nccl")
    backend = os.environ.get("BPG_BENCH_BACKEND", "nccl")
    coll_device = "cuda" if backend == "nccl" else "cpu"
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an AMD GPU: the product has no CPU path")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            if world > ndev:
                raise SystemExit("bench.py: %d ranks but %d GPUs; one rank per GPU over RCCL (set BPG_BENCH_BACKEND=gloo to rehearse on fewer GPUs)" % (world, ndev))
            device_index = local_rank
        else:
            device_index = local_rank % ndev
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == backend
    else:
        dist = None
        device_index = 0
    placement = pin_near_gpu(torch, device_index)          # before the library creates its threads
    # a chain thread needs a core to itself: plan this rank's host threads for the cores it really has (NUMA node shared with the other ranks
    # placed there, cgroup quota) - cores_for_rank / plan_chain_pool above, unit-tested on synthetic topologies in tests/test_host_logic.py
    try:
        cores_here = placement.get("cores_allowed") or max(1, len(os.sched_getaffinity(0)) // 2)
        sharing = 1
        if dist is not None:
            nodes = [None] * world
            dist.all_gather_object(nodes, placement.get("numa_node"))
            sharing = max(1, sum(1 for x in nodes if x == placement.get("numa_node"))) if placement.get("pinned") else world
        quota = cpu_quota()
        per_rank = cores_for_rank(cores_here, sharing, None if quota is None else quota * sharing // max(1, world))
        fit = max(2, per_rank - 2)
        if args.chain_workers > fit:
            log("rank %d: %d cores for %d rank(s) here (quota %s): --chain-workers %d -> %d" % (rank, cores_here, sharing, quota, args.chain_workers, fit))
            args.chain_workers = fit
            if not args.chain_pool:
                args.streams = min(args.streams, fit)
        placement["cores_per_rank"] = per_rank
        placement["cpu_quota"] = quota
    except Exception as e:      # noqa: BLE001 - sizing is an optimisation
        placement["sizing_error"] = repr(e)
        per_rank = 16

    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import workloads
    from bulletproofs_gadgets_amd.batch import gather_proofs, shard_indices
    ctx = bpg.Context(device_index, **ENGINE)
    n_streams = max(1, min(args.streams, args.steps))         # more streams than steps would only allocate
    pool_lanes = []
    if args.chain_pool == "auto":
        pool_lanes = plan_chain_pool(per_rank, n_streams)
    else:
        for part in filter(None, args.chain_pool.split(",")):
            cnt, _, ln = part.partition("x")
            pool_lanes += [int(ln or 1)] * int(cnt)
    pool = bpg.ChainPool(pool_lanes) if (pool_lanes and not args.no_prefetch) else None
    lane_workers = max(1, -(-max(1, args.chain_workers) // n_streams))       # chain threads per proving stream
    ctx.set_chain_workers(lane_workers)
    lanes_per_thread = max(1, min(8, args.chain_lanes))
    if lanes_per_thread > 1:
        ctx.set_chain_lanes(lanes_per_thread)
    t0 = time.perf_counter()
    a = workloads.merkle_full_tree(ctx, leaves=args.leaves, seed=(args.leaf_seed if rank == 0 else (args.leaf_seed or 0) + rank) if (rank or args.leaf_seed is not None) else None)
    inst = a.prover.instance()
    state = a.transcript.state
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.gens_ensure(a.gens_capacity)
    t_gens = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = ctx.upload(inst)
    t_up = time.perf_counter() - t0
    if rank == 0:
        log("workload: full %d-leaf MiMC Merkle tree n=%d N=%d q=%d m=%d | assembly+commit %.2fs gens %.2fs upload %.2fs"
            % (args.leaves, inst.n, a.gens_capacity, inst.q, inst.m, t_asm, t_gens, t_up))
    proof_len = bpg.lib().bpg_proof_size(inst.n, 0)
    prefetch = not args.no_prefetch
    extra_lanes = []
    for _ in range(n_streams - 1):
        c2 = bpg.Context(device_index, **ENGINE)
        c2.set_chain_workers(lane_workers)
        if lanes_per_thread > 1:
            c2.set_chain_lanes(lanes_per_thread)
        c2.gens_ensure(a.gens_capacity)
        extra_lanes.append((c2, c2.upload(inst)))

    def seed_for(step):
        return bytes([rank & 0xff, step & 0xff, (step >> 8) & 0xff]) + bytes(29)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if pool is not None:
        for c_ in [ctx] + [c2 for c2, _ in extra_lanes]:
            c_.attach_chain_pool(pool, 2)                     # the chain of a stream's next step is drawn while it proves the current one
    lanes = [(ctx, res)]                                     # proving streams of this rank: (context, resident circuit)
    done_at = []                                             # completion time of every proof of the current region (list.append is atomic)

    def prove_steps(seeds):
        """The timed steps: seeds[p::P] go to proving stream p (own thread, context, HIP stream, chain workers); P = 1 is prove_sequence.
        With several ranks the finished proofs of all steps are gathered once, at the end (inside the timed region)."""
        P = len(lanes)
        if P == 1:
            outs = prove_sequence(seeds, gather_each=False)
        else:
            import threading
            outs, errs = [None] * len(seeds), []

            def work(p):
                try:
                    for k, o in enumerate(prove_sequence(seeds[p::P], gather_each=False, lane=p)):
                        outs[p + k * P] = o
                except Exception as e:      # noqa: BLE001
                    errs.append(repr(e))
            th = [threading.Thread(target=work, args=(p,)) for p in range(P)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise RuntimeError(errs[0])
        if dist is not None:     # the only data that crosses xGMI: the finished proof bytes (one RCCL all_gather for the steps of the region)
            proofs = gather_proofs({rank + world * i: o[0] for i, o in enumerate(outs)}, world * len(seeds), proof_len, dist, device=coll_device)
            assert len(proofs) == world * len(seeds)
        return outs

    def prove_sequence(seeds, gather_each=True, ahead=None, lane=0):
        """Independent proofs, one after the other on this rank's context (one proving thread, one HIP stream).  With prefetch the chains of
        the next `ahead` proofs are queued on the chain worker before proof i is proved (bpg_blinding_begin: same bytes as drawing them inside
        prove); every chain of the sequence starts and ends inside it."""
        c_, r_ = lanes[lane]
        ahead = (1 if pool is not None else lane_workers * lanes_per_thread) if ahead is None else ahead
        outs = []
        queued = 0
        for i, s in enumerate(seeds):
            while prefetch and queued < len(seeds) and queued <= i + ahead:                  # streams i .. i + ahead alive (workers + 1 at most)
                c_.blinding_begin(state, inst.v_blinding, seeds[queued], inst.n)
                queued += 1
            out = r_.prove(state, inst.v_blinding, s, 0)
            done_at.append(time.perf_counter())
            if dist is not None and gather_each:   # the only data that crosses xGMI: the finished proof bytes (one RCCL all_gather per step)
                proofs = gather_proofs({rank: out[0]}, world, proof_len, dist, device=coll_device)
                assert len(proofs) == world
            outs.append(out)
        return outs

    def allreduce(value, op):
        if dist is None:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def gather_objects(obj):
        if dist is None:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    def collect_profile():
        prof = ctx.profile_report()
        ctx.profile_set(0)
        return prof

    if args.in_flight_only:
        res.prove(state, inst.v_blinding, seed_for(999), 0)
        out = in_flight_throughput(bpg, ctx, res, inst, state, a.gens_capacity, device_index, max(args.in_flight, 2), max(2, args.in_flight_steps),
                                   workers=max(1, args.in_flight_workers), lanes=max(1, min(8, args.in_flight_lanes)))
        if rank == 0:
            print(json.dumps({"in_flight": out, "schedule": ctx.schedule(), "leaf_seed": args.leaf_seed}), flush=True)
        return

    # ---- warm-up (not steps: the first proof of a context sizes its device workspaces), then EXACTLY `steps` timed proofs
    lanes.extend(extra_lanes)
    prove_steps([seed_for(2000 + i) for i in range(max(args.warmup, len(lanes)))])
    for c_, _ in lanes:
        c_.profile_set(1)       # HIP events around the bucket sweep and the generator folds only (11-13 launches per proof)
    barrier()
    del done_at[:]
    cpu0 = time.process_time()                               # CPU time of every thread of this process (chain pool and proving threads included)
    t0 = time.perf_counter()
    outs = prove_steps([seed_for(i) for i in range(args.steps)])
    barrier()
    elapsed_local = time.perf_counter() - t0
    cpu_timed = time.process_time() - cpu0
    # the same timed steps seen from their completion times (rank 0): the region opens with every chain still to be drawn, so nothing can
    # finish for ~0.3 s; what the GPU does meanwhile (A_I, A_O, S under the chains) and how the completions bunch afterwards is the timeline
    done = sorted(done_at)
    completions = None
    if done:
        completions = {"first_ms": (done[0] - t0) * 1e3, "median_ms": (done[len(done) // 2] - t0) * 1e3, "last_ms": (done[-1] - t0) * 1e3,
                       "note": "rank 0: when the first, the median and the last of the timed proofs finished, from the start of the timed region"}
    prof = {}
    for c_, _ in lanes:          # the streams' HIP-event records add up
        rep = c_.profile_report()
        c_.profile_set(0)
        for name, v in rep.items():
            acc = prof.setdefault(name, {k: 0 for k in v})
            for k in v:
                acc[k] += v[k]
    try:
        free_b, total_b = torch.cuda.mem_get_info(device_index)
        hbm_used = {"in_use_GB": (total_b - free_b) / 1e9, "total_GB": total_b / 1e9,
                    "note": "device memory in use on this rank's GPU with all proving streams alive (generator tables and odd multiples once per device, workspaces per stream)",
                    "model": memory_model(inst.n, a.gens_capacity, inst.q, inst.m, args.profile, n_streams, nnz=inst.nnz),
                    "model_note": "bench.memory_model: csrc/engine.hip's allocations restated - what `--plan-only` quotes per rank"}
    except Exception:       # noqa: BLE001
        hbm_used = None
    del lanes[1:]                # the other legs use the first stream only
    for c2, r2 in extra_lanes:
        r2.free(); c2.close()
    ctx.set_chain_workers(max(1, args.chain_workers))       # (detaches from the pool: the other legs use the context's own chain worker)
    elapsed = allreduce(elapsed_local, dist.ReduceOp.MAX if dist else None)
    q_total = allreduce(float(inst.q), dist.ReduceOp.SUM if dist else None)
    chain_cpu = ctx.chain_cpu()
    ranks_seen = gather_objects({"rank": rank, "device": device_index, "pci": placement.get("pci"), "numa_node": placement.get("numa_node"),
                                 "cpus_allowed": placement.get("cpus_allowed"), "chain_cpu": chain_cpu, "main_cpu": current_cpu(),
                                 "ms_per_step": elapsed_local / args.steps * 1e3, "proving_streams": n_streams, "chain_pool_lanes": pool_lanes if pool is not None else None,
                                 "hbm_in_use_GB": hbm_used["in_use_GB"] if hbm_used else None})      # the DEVICE's figure: ranks that share a card (gloo rehearsal) see the sum
    last = outs[-1]

    # ---- strong-scaling leg: a fixed batch of independent proofs, sharded round-robin over the ranks (north_star: 8 proofs, >= 6x at 8 GPUs)
    batch_info = None
    if args.batch > 0 and not args.headline_only:
        mine = shard_indices(args.batch, rank, world)
        ctx.set_chain_workers(1)
        barrier()
        t0 = time.perf_counter()
        bouts = prove_sequence([seed_for(3000 + i) for i in mine], gather_each=False, ahead=1)      # one chain at a time per rank: the reference's one process per proof
        local = {i: o[0] for i, o in zip(mine, bouts)}
        if dist is not None:
            allp = gather_proofs(local, args.batch, proof_len, dist, device=coll_device)
        else:
            allp = [local[i] for i in range(args.batch)]
        barrier()
        dtb = allreduce(time.perf_counter() - t0, dist.ReduceOp.MAX if dist else None)
        assert len(allp) == args.batch and all(len(p) == proof_len for p in allp)
        ctx.set_chain_workers(max(1, args.chain_workers))
        barrier()
        t0 = time.perf_counter()
        prove_sequence([seed_for(3500 + i) for i in mine], gather_each=False, ahead=max(1, args.chain_workers))
        barrier()
        dtb_deep = allreduce(time.perf_counter() - t0, dist.ReduceOp.MAX if dist else None)
        batch_info = {"proofs": args.batch, "ranks": world, "proofs_per_rank": len(shard_indices(args.batch, 0, world)), "seconds": dtb,
                      "seconds_with_all_chain_workers": dtb_deep,
                      "value": float(inst.q) * args.batch / dtb, "unit": "constraints/s", "scaling": "strong",
                      "note": "fixed batch of independent 2^20 proofs, proof i on rank i mod N. `seconds`: one chain at a time per rank (the reference: one prover "
                              "process per proof, src/bin/prover.rs:47-100) - the strong-scaling leg, speed-up = seconds at N=1 / seconds at N; "
                              "`seconds_with_all_chain_workers`: the same batch with every rank drawing --chain-workers chains side by side"}
        # what N = 8 can reach on this batch: each rank then proves ONE proof, which costs one chain + one proof's kernels (single_proof_latency_ms,
        # filled in below on rank 0).  Two ratios, because there are two N = 1 baselines: one chain at a time (how the reference would run the
        # batch: one prover process after the other) and this library's own best on one GPU (all chain workers).  north_star's ">= 6x at 8 GPUs
        # on an 8-proof batch" holds against the first and cannot hold against the second: one GPU with a dozen host cores already overlaps the
        # eight chains, so seven more GPUs only remove the ~0.2 s of kernels that followed them

    # ---- N > 1: the saturated figure on every GPU at once (secondary, as at N = 1): each rank keeps min(in-flight, 8) independent proofs in
    # flight on its own card; the sustained rates add up (no collective inside the leg, one all-reduce of the rates after it)
    multi_thr = None
    if dist is not None and args.in_flight > 1 and not args.headline_only:
        n_if = min(args.in_flight, 8)
        rate, err = 0.0, None
        try:
            barrier()
            t = in_flight_throughput(bpg, ctx, res, inst, state, a.gens_capacity, device_index, n_if, max(2, args.in_flight_steps // 2),
                                     workers=max(1, args.in_flight_workers), lanes=max(1, min(8, args.in_flight_lanes)), restore_workers=args.chain_workers)
            rate = 1e3 / t["ms_per_proof"]
        except Exception as e:      # noqa: BLE001 - secondary: a rank that fails contributes nothing, the collectives below still match
            err = repr(e)
        total = allreduce(rate, dist.ReduceOp.SUM)
        slowest = allreduce(rate if rate > 0 else 1e30, dist.ReduceOp.MIN)
        failed = allreduce(0.0 if err is None else 1.0, dist.ReduceOp.SUM)
        multi_thr = {"proofs_in_flight_per_gpu": n_if, "value": float(inst.q) * total, "unit": "constraints/s", "proofs_per_s": total,
                     "slowest_rank_proofs_per_s": slowest if slowest < 1e29 else 0.0, "ranks_failed": int(failed), "error_rank0": err,
                     "note": "sum over the ranks of each GPU's sustained rate (proving streams per GPU as given, one chain thread with --in-flight-lanes lanes "
                             "each); not the headline"}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- rank 0: untimed diagnostics and secondary figures
    tm = kernels = None
    latency_ms = None
    gpu_ms_per_proof = None
    lone_proof = None
    if not args.headline_only or args.lone_proof:
        # one proof alone, its chain drawn inside the call (what a single cold request costs), with phase timings
        t0 = time.perf_counter()
        alone = res.prove(state, inst.v_blinding, seed_for(0), 0, timings=True)
        latency_ms = (time.perf_counter() - t0) * 1e3
        tm = alone[2]
        if alone[0] != outs[0][0]:
            raise RuntimeError("the pipelined step and the stand-alone proof of the same seed differ")
        ctx.profile_set(2)
        res.prove(state, inst.v_blinding, seed_for(5001), 0)
        kernels = collect_profile()
        gpu_ms_per_proof = sum(v["total_ms"] for v in kernels.values())
        kern_only = {k: v for k, v in kernels.items() if not k.startswith("_")}
        lone_proof = {"profile": args.profile, "phase_ms": tm, "wall_ms": latency_ms, "kernel_ms_sum": sum(v["total_ms"] for v in kern_only.values()),
                      "launches": int(sum(v["count"] for v in kern_only.values())),
                      "note": "one 2^20 proof ALONE on the device, its chain drawn inside the call: phase times of the library's clock (rng_host = the 0.30 s chain, under "
                              "which A_I, A_O and most of S run), the sum of the HIP-event durations of every kernel of another lone proof, and the launches of that proof"}

    single, prof_isolated = None, None
    if not args.headline_only and prefetch and args.chain_workers > 1:
        # the round-1 / verdict figure for continuity: the same sequence with ONE chain drawn at a time (the chain of step i+1 under the kernels of step i)
        k1 = min(4, args.steps)
        ctx.set_chain_workers(1)                             # ONE host thread draws: queued chains run one after the other
        ctx.profile_set(1)                                   # and ONE HIP stream: the kernels run undisturbed - the isolated figures of `roofline`
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s1 = prove_sequence([seed_for(i) for i in range(k1)], gather_each=False, ahead=1)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / k1
        prof_isolated = collect_profile()
        ctx.set_chain_workers(max(1, args.chain_workers))
        assert [o[0] for o in s1] == [o[0] for o in outs[:k1]], "the same seeds gave other bytes with one chain at a time"
        single = {"steps": k1, "ms_per_step": dt1 * 1e3, "value": inst.q / dt1, "unit": "constraints/s",
                  "note": "one host thread draws the chains, one at a time (round 1 measured 340 ms per step without any prefetch); same proof bytes as the timed steps"}

    # ONE host thread for the chains of a whole sequence: bpg_ctx_set_chain_lanes(8) puts the sponges of eight queued proofs into the lanes of ZMM
    # registers (6x the chain throughput of a core); one proving stream, so 2 host threads per GPU in all
    one_thread = None
    if not args.headline_only and prefetch and world == 1:
        try:
            k8 = 16
            ctx.set_chain_workers(1); ctx.set_chain_lanes(8)
            seeds8 = [seed_for(9000 + i) for i in range(k8)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s8 = prove_sequence(seeds8, gather_each=False, ahead=7)
            torch.cuda.synchronize()
            dt8 = (time.perf_counter() - t0) / k8
            ctx.set_chain_lanes(1)
            alone8 = res.prove(state, inst.v_blinding, seeds8[3], 0)[0]
            assert s8[3][0] == alone8, "a proof of the lockstep sequence differs from the stand-alone proof of the same seed"
            one_thread = {"steps": k8, "ms_per_step": dt8 * 1e3, "value": inst.q / dt8, "unit": "constraints/s", "chain_threads": 1, "chain_lanes": 8, "proving_streams": 1,
                          "note": "one chain thread draws the chains of eight proofs in lockstep (eight sponges in the 64-bit lanes of ZMM registers: 193 ns per "
                                  "draw of all eight against 152 ns for one), one proving stream; every chain starts inside the timed sequence; same proof bytes"}
        except Exception as e:      # noqa: BLE001 - secondary
            one_thread = {"error": repr(e)}
        finally:
            ctx.set_chain_lanes(1)
            ctx.set_chain_workers(max(1, args.chain_workers))

    verify_info = None
    if not args.headline_only:
        coms = b"".join(a.commitments)
        t0 = time.perf_counter()
        rcs = [res.verify(state, coms, last[0]) for _ in range(3)]
        dtv = (time.perf_counter() - t0) / 3
        bad = bytearray(last[0]); bad[70] ^= 1
        verify_info = {"ms": dtv * 1e3, "accepted": all(r == 0 for r in rcs), "tampered_rejected": res.verify(state, coms, bytes(bad)) != 0,
                       "note": "bpg_r1cs_verify_resident: transcript replay on the host + one (2N+m+2lgN+13)-term MSM; not the headline"}
    expanded = None
    if not args.headline_only:
        res.prove(state, inst.v_blinding, seed_for(7000), bpg.FLAG_EXPANDED_BLINDING)
        t0 = time.perf_counter()
        for i in range(3):
            pe = res.prove(state, inst.v_blinding, seed_for(7001 + i), bpg.FLAG_EXPANDED_BLINDING)
        dte = (time.perf_counter() - t0) / 3
        expanded = {"ms_per_proof": dte * 1e3, "value": inst.q / dte, "unit": "constraints/s", "verified": res.verify(state, b"".join(a.commitments), pe[0]) == 0,
                    "note": "BPG_FLAG_EXPANDED_BLINDING (include/bpg.h): opt-in, not upstream's blinding derivation; not the headline"}

    # ---- roofline of the dominant kernel (HIP events on the engine's stream inside the timed steps) and of the whole proof
    src = source_hash()
    traffic_tab, traffic_note = {}, "no PMC profile stored"
    tfile = ROOT / "profiles" / "pmc_traffic.json"
    if tfile.exists():
        try:
            stored = json.loads(tfile.read_text())
            if stored.get("_meta", {}).get("source_hash") == src:
                traffic_tab = stored
                traffic_note = "rocprofv3 --pmc passes of these sources (profiles/pmc_traffic.json: %s)" % stored["_meta"].get("note", "")
            else:
                traffic_note = "profiles/pmc_traffic.json was taken from other sources (hash %s, these are %s): not quoted" % (stored.get("_meta", {}).get("source_hash"), src)
        except Exception as e:      # noqa: BLE001
            traffic_note = "profiles/pmc_traffic.json unreadable: %r" % (e,)
    peak_fm = ctx.bench_fe_mul(2000)
    # the schedule the LIBRARY settled on (bpg_profile_report "_schedule"), not a re-reading of the environment: tail start, fold groups, window caps
    schedule = ctx.schedule()
    fold_group, tt_lg = int(schedule["fold_group"]), int(schedule["tt_lg"])
    if a.gens_capacity <= (1 << int(schedule["tt_orig_lg"])):
        tt_lg = a.gens_capacity.bit_length() - 1              # the original generators are frozen at round 0: no sweep above the tail
    first_group = min(fold_group, max(a.gens_capacity.bit_length() - 1 - tt_lg, 0))
    rocprof_name = {"k_fold_points_reg": "k_fold_points_reg<%d>" % ((1 << first_group) - 1)}

    def kernel_roofline(name, table=None):
        k = (table or prof)[name]
        secs = k["total_ms"] * 1e-3
        launches = max(k["count"], 1)
        t = traffic_tab.get(rocprof_name.get(name, name))
        return {"kernel": rocprof_name.get(name, name), "launches": k["count"], "avg_launch_ms": k["total_ms"] / launches,
                "alg_bytes_per_launch": k["alg_bytes"] / launches, "achieved": k["alg_bytes"] / secs / 1e9 if secs > 0 else 0.0,
                "frac": k["alg_bytes"] / secs / 8e12 if secs > 0 else 0.0,
                "traffic": t["hbm_bytes_per_launch"] if t else None,
                "traffic_useful_fetch": t.get("useful_fetch_bytes_per_launch") if t else None,
                "device_GBps": k["device_bytes"] / secs / 1e9 if secs > 0 else 0.0,
                "valu": {"unit": "field-mult/s", "achieved": k["field_mults"] / secs if secs > 0 else 0.0, "peak": peak_fm,
                         "frac": (k["field_mults"] / secs / peak_fm) if secs > 0 and peak_fm > 0 else 0.0,
                         "peak_source": "k_bench_fe_mul microbenchmark on this device"}}
    ranked = sorted((n for n in prof if prof[n]["count"]), key=lambda n: -prof[n]["total_ms"])
    # a tie (within 3 %) goes to the kernel with more launches, so that the line names the same kernel on every run
    if len(ranked) > 1 and prof[ranked[1]]["total_ms"] > 0.97 * prof[ranked[0]]["total_ms"] and prof[ranked[1]]["count"] > prof[ranked[0]]["count"]:
        ranked[0], ranked[1] = ranked[1], ranked[0]
    b_alg = 576.0 * inst.n + 448.0 * a.gens_capacity           # algorithmic bytes of one proof
    t_step = elapsed / args.steps
    whole = {"alg_bytes": b_alg, "formula": "576 n + 448 N (SURVEY.md 8d)", "seconds": t_step, "achieved": b_alg / t_step / 1e9, "unit": "GB/s", "frac": b_alg / t_step / 8e12}
    if ranked:
        dom = kernel_roofline(ranked[0])
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": 8000.0, "unit": "GB/s",
                    "frac": dom["frac"], "traffic": dom["traffic"], "traffic_useful_fetch": dom["traffic_useful_fetch"], "traffic_source": traffic_note, "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"],
                    "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                    "traffic_definition": "HBM-side bytes per launch from the PMC counters: FETCH_SIZE x 2 (gfx950 tallies 64 B per 128-byte request) + WRITE_SIZE; "
                                          "traffic_useful_fetch = FETCH_SIZE x the factor calibrated on random 96-byte rows (the bytes the kernel asked for, not HBM traffic)",
                    "alg_bytes_definition": "information content, SURVEY.md 8(d): S + P = 64 B per MSM term for the bucket sweep (once per term, not per window); "
                                            "32 B per point read or written for a generator fold",
                    "device_GBps": dom["device_GBps"],
                    "note": "integer-VALU bound path (255-bit modular arithmetic): the HBM fraction is reported as required, the binding roofline is 'valu'; "
                            "kernel names as rocprofv3 prints them (profiles/*_kernel_stats.csv)",
                    "valu": dom["valu"], "whole_proof": whole, "other_kernels": [kernel_roofline(n) for n in ranked[1:]]}
        # `iso_name` = the kernel that does a kernel's job in the single-stream leg (the same name, unless a kernel has a shared-device shape of its own)
        def iso_name(n):
            if not prof_isolated:
                return None
            if n in prof_isolated:
                return n
            return next((x for x in SWEEPS if n in SWEEPS and x in prof_isolated), None)
        if n_streams > 1 and iso_name(ranked[0]):
            # with several proving streams a launch inside the timed steps shares the CUs with kernels of other proofs, and its duration says how
            # the GPU was shared, not how good the kernel is (14 streams: 2.4 - 3.1 ms per sweep from run to run, 0.97 ms alone).  The headline
            # figures of `roofline` are therefore the kernel ALONE on the GPU - HIP events in the single-stream leg of this same run, the
            # command whose rocprofv3 --stats summary is profiles/*_kernel_stats_single_stream.csv - and the shared figures follow beside them
            shared = {k: roofline[k] for k in ("launches", "avg_launch_ms", "alg_bytes_per_launch", "achieved", "frac", "device_GBps", "valu", "other_kernels")}
            shared["note"] = ("the same kernel inside the timed steps, where %d proving streams share the GPU (HIP events on each stream, summed); "
                              "rocprofv3 --stats of the headline command: profiles/*_rocprof_kernel_stats.csv" % n_streams)
            iso = kernel_roofline(iso_name(ranked[0]), prof_isolated)
            for k in ("launches", "avg_launch_ms", "alg_bytes_per_launch", "achieved", "frac", "device_GBps", "valu"):
                roofline[k] = iso[k]
            if iso["kernel"] != roofline["kernel"]:
                roofline["traffic"], roofline["traffic_useful_fetch"] = iso["traffic"], iso["traffic_useful_fetch"]      # the counter passes profile one stream alone
                roofline["kernel_alone"] = iso["kernel"]
                roofline["note"] += "; alone on the device this kernel's job is done by %s" % iso["kernel"]
            roofline["other_kernels"] = [kernel_roofline(n, prof_isolated) for n in ranked[1:] if n in prof_isolated]
            roofline["measured"] = ("kernel alone on the GPU: HIP events on the engine's stream in the single-stream leg of this run (one proving stream, one "
                                    "chain thread; rocprofv3 --stats of that command: profiles/*_kernel_stats_single_stream.csv). `timed_steps_shared` = the "
                                    "same kernel inside the timed steps of the headline")
            roofline["timed_steps_shared"] = shared
        elif n_streams > 1:
            roofline["note"] += ("; %d proving streams share the GPU inside the timed steps, so a launch there overlaps kernels of another proof and lasts "
                                 "longer than it does alone (run without --headline-only for the kernel alone)" % n_streams)
    else:
        roofline = {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0, "traffic": None, "whole_proof": whole,
                    "note": "no fold / bucket-sweep launch in the timed steps (table-driven schedule at this size)"}
    if ranked and roofline.get("kernel") in SWEEPS:
        # both term counts of the bucket sweep: what the launches process - the rounds of a fold group run on the group-start tables
        # with expanded scalars, so rounds 2 and 3 of a group visit as many terms as round 1 - and what the survey's round sizes add up to
        # (A_I, A_O, S: 5n terms; L_k, R_k of round k: 2 N_k terms for the rounds above the table-driven tail)
        N, k = a.gens_capacity, 0
        survey_terms = 5 * inst.n
        while (tt_lg == 0 or (N >> k) > (1 << tt_lg)) and (N >> k) > 1:
            survey_terms += 2 * (N >> k); k += 1
        iso = n_streams > 1 and prof_isolated and any(x in prof_isolated for x in SWEEPS)
        src_prof = prof_isolated if iso else prof
        sweeps = src_prof[next(x for x in SWEEPS if x in src_prof)]
        proofs_seen = single["steps"] if iso else args.steps
        processed = sweeps["alg_bytes"] / 64.0
        secs = sweeps["total_ms"] * 1e-3
        roofline["alg_terms_processed"] = processed / proofs_seen
        roofline["alg_terms_survey"] = survey_terms
        roofline["frac_survey_terms"] = (64.0 * survey_terms * proofs_seen / secs / 8e12) if secs > 0 else 0.0
        roofline["terms_note"] = ("`frac` counts 64 B for every term a launch processes (alg_terms_processed per proof); `frac_survey_terms` counts the survey's "
                                  "round sizes only (alg_terms_survey per proof) over the same launch time; per-proof figures assume %d proofs in the profiled "
                                  "sequence" % proofs_seen)
    fm_per_proof = sum(prof[n]["field_mults"] for n in prof) / max(args.steps, 1)

    out = {"metric": METRIC, "value": q_total * args.steps / elapsed,
           "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": t_step * 1e3, "gates_per_s": float(inst.n) * world * args.steps / elapsed, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "dtype_note": "8 x u32 limbs: 255-bit modular integer arithmetic (v_mad_u64_u32), no floating point", "data": "synthetic",
           "config": {"workload": "full %d-leaf MiMC Merkle tree (reference merkle_tree_gadget.rs:473-545); a step = one independent proof (own seed) per GPU; the %d "
                                  "timed steps run CONCURRENTLY on %d proving streams per GPU and ms_per_step = timed region / %d - a rate, not the latency of a "
                                  "proof (that is single_proof_latency_ms; the sustained rate of a long sequence is throughput.ms_per_proof)"
                                  % (args.leaves, args.steps, n_streams, args.steps), "n": inst.n, "N": a.gens_capacity, "q": inst.q, "m": inst.m,
                      "inputs": "flattened R1CS instance + generator tables resident in HBM", "rng": "Merlin TranscriptRng (upstream-exact)",
                      "chain": (("the steps are dealt round-robin to %d proving streams per GPU (engine contexts: own HIP stream and proving thread, generator tables "
                                 "shared); the serial TranscriptRng chains (one per proof, 2n+8 dependent Keccak-f, upstream-exact) are drawn by ONE pool of %d chain "
                                 "threads shared by the streams (bpg_chain_pool_create; lanes per thread %s: a thread with one lane draws a chain in 0.30 s, a lockstep "
                                 "thread its chains in 0.38 s), sized for the %s cores this rank has, so that the first chain of every stream is drawn at once. All %d "
                                 "chains start and end inside the timed region: nothing can finish in its first ~0.3 s, during which the streams run A_I, A_O and "
                                 "most of S under their chains" % (n_streams, len(pool_lanes), pool_lanes, placement.get("cores_per_rank"), args.steps))
                                if pool is not None else
                                ("the steps are dealt round-robin to %d proving streams per GPU (engine contexts: own HIP stream and proving thread, generator tables "
                                 "shared); the serial TranscriptRng chains (one per proof, 2n+8 dependent Keccak-f, upstream-exact) are drawn by %d chain threads, "
                                 "%d per stream, each stream keeping the chain of its next step queued. All %d chains start and end inside the timed region: "
                                 "nothing can finish in its first ~0.3 s, during which the streams run A_I, A_O and most of S under their chains"
                                 % (n_streams, lane_workers * n_streams, lane_workers, args.steps)))
                               if prefetch else "every chain is drawn inside its own prove call",
                      "host_threads_per_gpu": {"chain_workers": (len(pool_lanes) if pool is not None else lane_workers * n_streams) if prefetch else 0, "proving": n_streams,
                                               "chain_pool_lanes": pool_lanes if pool is not None else None,
                                               "cpu_seconds_timed": cpu_timed, "busy_cores_avg": cpu_timed / max(elapsed_local, 1e-9),
                                               "cpu_seconds_of_chains_one_core_each": args.steps * (2 * inst.n + 8) * 152e-9,
                                               "note": "chain threads are busy for the whole chain; proving threads sleep in their stream waits (blocking sync). "
                                                       "cpu_seconds_timed = process CPU time (all threads, rank 0) over the timed steps, busy_cores_avg = that / wall; "
                                                       "cpu_seconds_of_chains_one_core_each = steps x (2n+8) draws x 152 ns: what the chains cost when each has a core to itself (a lockstep "
                                                       "thread draws its up to eight chains for 191 ns per draw of all of them)"},
                      "proving_streams_per_gpu": n_streams,
                      "backend": backend if world > 1 else None},
           "roofline": roofline, "ranks_seen": ranks_seen, "host": dict(host_description(), placement=placement),
           "single_stream": single, "one_host_thread": one_thread, "single_proof_latency_ms": latency_ms, "phase_ms": tm, "lone_proof": lone_proof, "verify": verify_info, "expanded_blinding": expanded,
           "setup_s": {"assembly_and_commit": t_asm, "generators": t_gens, "upload": t_up}, "source_hash": src, "schedule": schedule, "profile": args.profile}
    if completions is not None:
        out["completions"] = completions
    if hbm_used is not None:
        out["hbm_in_use"] = hbm_used
    if gpu_ms_per_proof is not None:
        out["gpu_busy"] = {"kernel_ms_per_proof": gpu_ms_per_proof, "fraction_of_step": gpu_ms_per_proof / (t_step * 1e3),
                           "note": "sum of the HIP-event durations of every kernel of one (untimed) proof / ms_per_step"}
    if batch_info is not None:
        if world == 1 and latency_ms:
            one = latency_ms * 1e-3
            batch_info["speedup_bound_at_8"] = {"one_proof_alone_s": one, "vs_one_chain_at_a_time": batch_info["seconds"] / one,
                                                "vs_all_chain_workers": batch_info["seconds_with_all_chain_workers"] / one,
                                                "claim": "vs_one_chain_at_a_time is the reference-like baseline (one prover process per proof, src/bin/prover.rs:47-100) and the one "
                                                         "north_star's >= 6x refers to; vs_all_chain_workers is the honest ratio against this library's own best on ONE GPU - "
                                                         "an 8-proof batch is bound by the 0.3 s host chain of a proof, which one GPU's host cores already draw side by side"}
        out["batch"] = batch_info
    if kernels is not None and args.kernel_profile:
        out["kernel_ms"] = {k: round(v["total_ms"], 3) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])}
    if world == 1 and args.in_flight > 1 and not args.headline_only:
        try:
            thr = in_flight_throughput(bpg, ctx, res, inst, state, a.gens_capacity, device_index, args.in_flight, max(2, args.in_flight_steps),
                                       workers=max(1, args.in_flight_workers), lanes=max(1, min(8, args.in_flight_lanes)), restore_workers=args.chain_workers)
            thr["note"] = ("sustained rate of ONE GPU over a long sequence: %d proving streams (engine contexts), each with %d chain thread(s) that draw(s) up to %d "
                           "queued chains in lockstep (bpg_ctx_set_chain_lanes) - the chains are ready before their proofs start, so A_I, A_O, S go through one "
                           "multiscalar pass and the GPU is the bound; not the headline"
                           % (args.in_flight, max(1, args.in_flight_workers), max(1, min(8, args.in_flight_lanes))))
            if gpu_ms_per_proof:
                thr["kernel_ms_per_proof_alone"] = gpu_ms_per_proof
                thr["gpu_time_fraction"] = gpu_ms_per_proof / thr["ms_per_proof"]
            thr["valu"] = {"unit": "field-mult/s", "achieved": fm_per_proof * 1e3 / thr["ms_per_proof"], "peak": peak_fm,
                           "frac": fm_per_proof * 1e3 / thr["ms_per_proof"] / peak_fm if peak_fm else None,
                           "counted": "field multiplications of the bucket sweeps and the generator folds only (%.3g per proof)" % fm_per_proof}
            thr["hbm"] = {"achieved": b_alg * 1e3 / thr["ms_per_proof"] / 1e9, "unit": "GB/s", "frac": b_alg * 1e3 / thr["ms_per_proof"] / 8e12}
            # counter evidence for the concurrent mix: two rocprofv3 --pmc passes of `bench.py --in-flight-only` (tools/profile_round.sh pmcthr), quoted only
            # when they were taken from these sources
            tthr = ROOT / "profiles" / "pmc_traffic_throughput.json"
            if tthr.exists():
                try:
                    meta = json.loads(tthr.read_text()).get("_meta", {})
                    if meta.get("source_hash") == src and meta.get("hbm_bytes_per_proof"):
                        thr["hbm"]["traffic_bytes_per_proof"] = meta["hbm_bytes_per_proof"]
                        thr["hbm"]["traffic_GBps"] = meta["hbm_bytes_per_proof"] * 1e3 / thr["ms_per_proof"] / 1e9
                        thr["hbm"]["traffic_frac_of_8TBps"] = meta["hbm_bytes_per_proof"] * 1e3 / thr["ms_per_proof"] / 8e12
                        thr["hbm"]["traffic_source"] = "profiles/pmc_traffic_throughput.json: FETCH_SIZE x 2 + WRITE_SIZE over %d proofs of the concurrent mix" % meta.get("proofs_profiled", 0)
                    else:
                        thr["hbm"]["traffic_source"] = "profiles/pmc_traffic_throughput.json was taken from other sources: not quoted"
                except Exception as e:      # noqa: BLE001
                    thr["hbm"]["traffic_source"] = "profiles/pmc_traffic_throughput.json unreadable: %r" % (e,)
            sch = ctx.schedule()
            thr["merging"] = {"merged_groups": sch["merged_last"], "terms_merged_away": sch["merged_skipped_last"], "merge_ms_once_per_witness": sch["merge_ms_last"],
                              "note": "A_I and A_O of this instance after the equal-scalar merging (csrc/hip/k_merge.cuh): groups of equal scalars and the terms they stand for; "
                                      "grouped once per uploaded witness, outside the timed steps - `throughput_variants.merge_per_proof` has it inside every proof"}
            out["throughput"] = thr
            try:
                out["throughput_variants"] = throughput_variants(bpg, workloads, device_index, inst, state, a.gens_capacity, args.in_flight, max(2, args.in_flight_steps),
                                                                 max(1, args.in_flight_workers), max(1, min(8, args.in_flight_lanes)), args.leaves)
            except Exception as e:      # noqa: BLE001
                out["throughput_variants"] = {"error": repr(e)}
        except Exception as e:      # noqa: BLE001 - a failed secondary measurement must not lose the headline
            out["throughput"] = {"error": repr(e)}
    if multi_thr is not None:
        out["throughput"] = multi_thr
    if world == 1 and not args.headline_only and args.leaves == 512:
        try:
            out["end_to_end"] = end_to_end(bpg, workloads, ctx, a.gens_capacity, last[0], seed_for(args.steps - 1))
        except Exception as e:      # noqa: BLE001
            out["end_to_end"] = {"error": repr(e)}
    if world == 1 and not args.headline_only:
        try:
            out["small_configs"] = small_configs(bpg, workloads, device_index)
        except Exception as e:      # noqa: BLE001
            out["small_configs"] = {"error": repr(e)}
    if world == 1 and not args.headline_only and not args.no_one_shot_leg and args.profile == "serving":
        try:
            out["value_one_shot"] = one_shot_leg(args)
        except Exception as e:      # noqa: BLE001
            out["value_one_shot"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline and not args.headline_only:
        log("cpu_baseline: oracle on %d leaves (about %d s on one core) ..." % (args.baseline_leaves, 150 * args.baseline_leaves // 512 + 2))
        cb = cpu_baseline(ctx, bpg, workloads, args.baseline_leaves, {"leaves": args.leaves, "a": a, "inst": inst, "res": res})
        cb["host"] = "%s, %d logical CPUs visible; 1 used" % (out["host"]["cpu_model"], os.cpu_count() or 0)
        # the ratio against the raw port and against a CPU time halved for upstream's avx2_backend (Cargo.toml:20)
        cb["gpu_over_cpu"] = out["value"] / cb["value"]
        cb["gpu_over_cpu_avx2_adjusted"] = out["value"] / (2.0 * cb["value"])
        cb["gpu_side_host_threads"] = ((len(pool_lanes) if pool is not None else lane_workers * n_streams) if prefetch else 0) + n_streams
        if single is not None:
            cb["single_stream_over_cpu"] = single["value"] / cb["value"]      # one GPU + two host threads against one core
        out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.plan_only:
        topo = discover_topology(args.gpus) or synthetic_topology(args.gpus)
        print(json.dumps(plan_node(args.gpus, args.streams, args.steps if args.steps > 3 else 20, args.profile, topo)), flush=True)
        return
    if args.headline_only:
        args.no_cpu_baseline, args.in_flight = True, 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
