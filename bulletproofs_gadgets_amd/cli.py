"""File driver for the reference's CLI formats: all seven gadget lines and OR blocks.

    python -m bulletproofs_gadgets_amd.cli prover   NAME      reads NAME.gadgets/.inst/.wtns, writes NAME.coms/.proof
    python -m bulletproofs_gadgets_amd.cli verifier NAME      reads NAME.gadgets/.inst/.coms/.proof, prints true/false
    python -m bulletproofs_gadgets_amd.cli prover|verifier --batch FILE [--gpus N]
        FILE lists one NAME per line: a batch of independent proofs (the reference's own batch is its CI workflow, prover then verifier over twelve stems:
        .github/workflows/integration_tests.yml:19-58).  One process per GPU; rank r takes stems r, r + N, ... and runs the driver above on each with
        ONE engine context (generators and tables derived once per rank), writes its own .coms/.proof; the finished proof bytes (verifier: the
        verdicts) are gathered over RCCL and rank 0 prints one summary line per stem.  Same files as N = 1 and as one run per stem.

Mirrors reference src/bin/prover.rs:47-100 and src/bin/verifier.rs:46-101 : the transcript label is the NAME argument (prover.rs:49-52); witnesses are committed in .wtns order and
written as "C{id}-{k} = 0x.." (assignment_parser.rs:152-169,213-220); derived commitments as "D{line}-{sub}-{k}";
MERKLE hashes every W leaf through hash_witness (prover.rs:160-190) and every I leaf through mimc_hash (prover.rs:192-200);
generators: round_pow2(#multipliers) (prover.rs:43-45,92). The top-level shadow ProverBuffer of the reference (prover.rs:66-86) records every operation
and replays it into the real prover at the end, which is the identity on multiplier / constraint order: top-level lines go to the prover directly.
EQUALS / UNEQUAL / SET_MEMBER / LESS_THAN follow prover.rs:340-532 and verifier.rs:262-424; OR [ { .. } { .. } ] blocks (also nested)
follow prover.rs:202-238 / or_conjunction.rs:4-38 with a recording ConstraintBuffer per block (row f3).
"""
import hashlib
import os
import re
import sys

from . import (BoundsCheck, BulletproofGens, ConstraintBuffer, Context, Equality, or_conjunction, Inequality, LessThan, MerkleTree256, MimcHash256, Prover,
               SetMembership, Transcript, Verifier, be_to_scalar, be_to_scalars, commit, commit_single, mimc_hash, scalar_to_be, L)

_VAR = re.compile(r"^\s*([A-Za-z][0-9]+(?:-[0-9]+){0,2})\s*=\s*0[xX]([0-9a-fA-F]+)\s*$")
MAX_OR_NESTING = 64      # OR blocks may nest (reference src/bin/prover.rs:219-234); the file is untrusted input, so the recursion is bounded (csrc/cli_main.cpp: same bound)


def _read_vars(path):
    out = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            m = _VAR.match(line)
            if not m:
                raise ValueError("cannot parse %r in %s" % (line, path))
            out.append((m.group(1), bytes.fromhex(m.group(2))))
    return out


def round_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


class _Blindings:
    """Scalar::random(&mut thread_rng()) stand-in: os.urandom, or a SHAKE256 stream when a seed is given (tests)."""
    def __init__(self, seed=None):
        self.seed, self.ctr = seed, 0

    def next(self):
        if self.seed is None:
            raw = os.urandom(64)
        else:
            raw = hashlib.shake_256(self.seed + self.ctr.to_bytes(8, "little")).digest(64)
            self.ctr += 1
        return (int.from_bytes(raw, "little") % L).to_bytes(32, "little")

    def take(self, k):
        return [self.next() for _ in range(k)]


def parse_tree(text):
    """Tree syntax of gadget_grammar.lalrpop:54-79 -> (instance names, witness names, pattern string), left to right."""
    toks = re.findall(r"\(|\)|[WI][0-9]+", text)
    pos = 0
    inst, wit = [], []

    def node(depth=0):
        nonlocal pos
        if depth > 64 or pos >= len(toks):      # nesting bounded like the native driver and the library (Pattern::MAX_DEPTH)
            raise ValueError("malformed tree")
        t = toks[pos]; pos += 1
        if t == "(":
            l = node(depth + 1); r = node(depth + 1)
            if pos >= len(toks) or toks[pos] != ")":
                raise ValueError("malformed tree")
            pos += 1
            return "(%s %s)" % (l, r)
        if t[0] == "W":
            wit.append(t); return "W"
        if t[0] == "I":
            inst.append(t); return "I"
        raise ValueError("malformed tree")
    pat = node()
    if pos != len(toks) or not pat.startswith("("):
        raise ValueError("malformed tree")
    return inst, wit, pat


def _setup(g, p, scalars, blindings):
    """Gadget::setup (reference src/gadget.rs:18-38); a prover that makes its commitments outside the library brings its own (workloads._setup)"""
    return p.gadget_setup(g, scalars, blindings) if hasattr(p, "gadget_setup") else g.setup(p, scalars, blindings)


def prover(name, ctx=None, seed=None, rng_seed=None, flags=0, quiet=False, two_pass=None, prover_cls=None, assemble_only=False):
    """two_pass (default: on, BPG_CLI_TWO_PASS=0 turns it off): every commitment of the .gadgets file is made first (pass 1: Gadget::setup's
    derived witnesses and hash_witness's images depend on witness and instance bytes only), the TranscriptRng chain of prove() starts on the
    context's chain worker as soon as the transcript is final (Prover.start_blinding), and pass 2 assembles the constraints beside it, replaying
    the cached commitments.  Same blinding draws, same .coms lines, same transcript, same proof bytes as the reference's single pass.
    assemble_only: stop before prove() and return (prover, transcript) - the statement as prove() would receive it (bench.py times the proof of a file
    configuration alone; tests/golden/gen_big_proof_fixtures.py hands it to the oracle).  prover_cls: a Prover subclass (the fixture generator's, which
    takes its commitments from the oracle and needs no device)."""
    if prover_cls is None:
        ctx = ctx or Context(0)
    if two_pass is None:
        two_pass = os.environ.get("BPG_CLI_TWO_PASS", "1") != "0"
    if prover_cls is not None:
        two_pass = False                                                  # no device context: no blinding stream to start early
    rng_seed = rng_seed if rng_seed is not None else os.urandom(32)
    st = {"pass": 0, "cache": [], "pos": 0, "est": 0, "or": False}

    def cached(make):
        """a commitment step: performed in the single pass and in pass 1 (remembered there), replayed in pass 2"""
        if st["pass"] == 2:
            r = st["cache"][st["pos"]]; st["pos"] += 1
            return r
        r = make()
        if st["pass"] == 1:
            st["cache"].append(r)
        return r

    def assemble(g, cs, vars_, derived):
        if st["pass"] != 1:
            g.prove(cs, vars_, derived)
    rnd = _Blindings(seed)
    transcript = Transcript(name.encode())
    p = (prover_cls or Prover)(ctx, transcript)
    instance = dict(_read_vars(name + ".inst"))
    witness = {}
    coms_lines = []
    for wname, data in _read_vars(name + ".wtns"):                       # assignment_parser.rs:152-169
        scalars, coms, vars_ = commit(p, data, rnd.take((len(data) + 31) // 32 or 1))
        witness[wname] = (scalars, coms, vars_, data)
        for k, c in enumerate(coms):
            coms_lines.append("C%s-%d = 0x%s\n" % (wname[1:], k, c.hex()))

    def derived_lines(coms, index, sub):                                  # assignment_parser.rs:198-211
        if st["pass"] == 2:
            return
        for k, c in enumerate(coms):
            coms_lines.append("D%d-%d-%d = 0x%s\n" % (index, sub, k, c.hex()))

    def single(wname):
        w = witness[wname]
        assert len(w[0]) == 1, "witness var %s is longer than 32 bytes" % wname
        return w

    def lc_of(token):
        if token[0] == "W":
            return single(token)[2][0]
        data = instance[token]
        assert len(data) <= 32, "instance var %s is longer than 32 bytes" % token
        return be_to_scalar(data)

    def hash_witness(wn, index, sub, cs):                                 # prover.rs:160-190 -> (image scalar, image Variable)
        w = witness[wn]
        image = mimc_hash(w[3])
        _, image_com, image_var = cached(lambda: commit_single(p, scalar_to_be(image), rnd.next()))
        hg = MimcHash256(image_var)
        dcoms, derived = cached(lambda: _setup(hg, p, w[0], rnd.take(2)))
        assemble(hg, cs, w[2], derived)
        st["est"] += 972 * (len(w[0]) + 1)
        derived_lines([image_com] + dcoms, index, sub)
        return image, image_var

    with open(name + ".gadgets") as f:
        lines = [l.rstrip("\n") for l in f]

    def do_gadget(line, index, cs):
        """one gadget line: commitments always go to the real prover p, constraints to cs (p itself or an OR-block buffer)"""
        parts = line.split()
        op = parts[0]
        if op == "BOUND":                                                 # prover.rs:253-276
            w = single(parts[1])
            lo, hi = instance[parts[2]], instance[parts[3]]
            g = BoundsCheck(lo, hi)
            st["est"] += 16 * len(hi)
            dcoms, derived = cached(lambda: _setup(g, p, w[0], rnd.take(2)))
            assemble(g, cs, w[2], derived)
            derived_lines(dcoms, index, 0)
        elif op == "HASH":                                                # prover.rs:278-305
            g = MimcHash256(lc_of(parts[1]))
            w = witness[parts[2]]
            st["est"] += 972 * (len(w[0]) + 1)
            dcoms, derived = cached(lambda: _setup(g, p, w[0], rnd.take(2)))
            assemble(g, cs, w[2], derived)
            derived_lines(dcoms, index, 0)
        elif op == "MERKLE":                                              # prover.rs:307-339
            root = lc_of(parts[1])
            inst_names, wit_names, pattern = parse_tree(line.split(None, 2)[2])
            inst_lcs = [mimc_hash(instance[i]) for i in inst_names]
            wit_lcs = [hash_witness(wn, index, sub, cs)[1] for sub, wn in enumerate(wit_names)]
            st["est"] += 1944 * pattern.count("(")
            assemble(MerkleTree256(root, inst_lcs, wit_lcs, pattern), cs, [], [])
        elif op == "EQUALS":                                              # prover.rs:340-358 (grammar: W I | I W | W W)
            left, right = (parts[1], parts[2]) if parts[1][0] == "W" else (parts[2], parts[1])
            right_lcs = witness[right][2] if right[0] == "W" else be_to_scalars(instance[right])
            assemble(Equality(right_lcs), cs, witness[left][2], [])
        elif op == "LESS_THAN":                                           # prover.rs:360-382
            l, r = single(parts[1]), single(parts[2])
            g = LessThan(l[2][0], l[0][0], r[2][0], r[0][0])
            st["est"] += 379
            dcoms, derived = cached(lambda: _setup(g, p, [], rnd.take(2)))
            assemble(g, cs, [], derived)
            derived_lines(dcoms, index, 0)
        elif op == "UNEQUAL":                                             # prover.rs:384-418
            left, right = (parts[1], parts[2]) if parts[1][0] == "W" else (parts[2], parts[1])
            lw = witness[left]
            if right[0] == "W":
                rs_, rl = witness[right][0], witness[right][2]
            else:
                rs_ = be_to_scalars(instance[right]); rl = rs_
            g = Inequality(rl, rs_)
            st["est"] += 2 * len(lw[0]) + 1
            dcoms, derived = cached(lambda: _setup(g, p, lw[0], rnd.take(2 * len(lw[0]) + 1)))
            assemble(g, cs, lw[2], derived)
            derived_lines(dcoms, index, 0)
        elif op == "SET_MEMBER":                                          # prover.rs:420-532
            member, elems = parts[1], parts[2:]
            if member[0] == "W":
                m_scalars, m_lcs = witness[member][0], witness[member][2]
            else:
                m_scalars = be_to_scalars(instance[member]); m_lcs = m_scalars
            m_scalar, m_lc = m_scalars[0], m_lcs[0]
            hashing = len(m_scalars) > 1
            w_vars, w_scalars, i_lcs, i_scalars = [], [], [], []
            if not hashing:
                for e in elems:
                    if e[0] == "W":
                        if len(witness[e][2]) == 1:
                            w_scalars.append(witness[e][0][0]); w_vars.append(witness[e][2][0])
                        else:
                            hashing = True
                    else:
                        sc_ = be_to_scalars(instance[e])
                        if len(sc_) == 1:
                            i_scalars.append(sc_[0]); i_lcs.append(sc_[0])
                        else:
                            hashing = True
            if hashing:                                                   # elements longer than one scalar: compare MiMC images
                sub = 1
                if member[0] == "W":
                    m_scalar, m_lc = hash_witness(member, index, sub, cs); sub += 1
                else:
                    m_scalar = mimc_hash(instance[member]); m_lc = m_scalar
                w_vars, w_scalars, i_lcs, i_scalars = [], [], [], []
                for e in elems:
                    if e[0] == "W":
                        sc_, var = hash_witness(e, index, sub, cs); sub += 1
                        w_vars.append(var); w_scalars.append(sc_)
                    else:
                        h = mimc_hash(instance[e]); i_lcs.append(h); i_scalars.append(h)
            g = SetMembership(m_lc, m_scalar, i_lcs, i_scalars)
            st["est"] += 2 * (len(w_scalars) + len(i_scalars))
            dcoms, derived = cached(lambda: _setup(g, p, w_scalars, rnd.take(len(w_scalars) + len(i_scalars))))
            assemble(g, cs, w_vars, derived)
            derived_lines(dcoms, index, 0)
        else:
            raise ValueError("unknown gadget line: %r" % line)

    def run_block(i, cs, closing, depth=0):
        """lines from i on; closing = None at top level, "]" inside an OR block (prover.rs:75-84 and :219-234)"""
        if depth > MAX_OR_NESTING:
            raise ValueError("OR blocks nested deeper than %d" % MAX_OR_NESTING)
        while i < len(lines):
            line, index = lines[i], i
            i += 1
            parts = line.split()
            if not parts:
                continue
            op = parts[0]
            if closing is not None and op == closing:
                return i
            if st["pass"] == 1:                                           # commitments only: OR clauses are walked in file order, nothing is recorded
                if op == "OR":
                    st["or"] = True
                    i = run_block(i, cs, "]", depth + 1)
                elif op not in ("}", "[", "{"):
                    do_gadget(line, index, cs)
                continue
            if op == "}":
                cs.rewind()                                               # end of a clause
            elif op == "OR":
                child = ConstraintBuffer(cs, True)
                i = run_block(i, child, "]", depth + 1)
                or_conjunction(cs, child)
            elif op in ("[", "{"):
                pass
            else:
                do_gadget(line, index, cs)
        if closing is not None:
            raise ValueError("unexpected end of input")
        return i
    if two_pass and hasattr(p, "start_blinding"):
        st["pass"] = 1
        run_block(0, p, None)
        est = st["est"] + 64
        p.start_blinding(rng_seed, 2 * est + 65536 if st["or"] else est)  # sizes the pinned buffer; a stream that is too short is simply not used
        st["pass"] = 2
        run_block(0, p, None)
        assert st["pos"] == len(st["cache"]), "pass 2 did not use every commitment of pass 1"
    else:
        run_block(0, p, None)
    if not quiet:
        print(p.num_constraints())                                        # prover.rs:89
    if assemble_only:
        return p, transcript
    cap = round_pow2(p.get_num_multiplications())
    proof = p.prove(BulletproofGens(ctx, cap), rng_seed, flags)
    with open(name + ".coms", "w") as f:
        f.writelines(coms_lines)
    with open(name + ".proof", "wb") as f:
        f.write(proof)
    return p, proof


def assemble_verifier(name):
    """Verifier-side replay (verifier.rs:46-90) up to, not including, verify(). Returns (Verifier, transcript)."""
    transcript = Transcript(name.encode())
    v = Verifier(transcript)
    instance = dict(_read_vars(name + ".inst"))
    commitments = {}
    for cname, data in _read_vars(name + ".coms"):                        # parse_coms: every line, file order
        if len(data) != 32:
            raise ValueError("commitment %s is not 32 bytes" % cname)
        commitments[cname] = v.commit(data)

    def all_commitments(wname):
        out, k = [], 0
        while "C%s-%d" % (wname[1:], k) in commitments:
            out.append(commitments["C%s-%d" % (wname[1:], k)]); k += 1
        if not out:
            raise KeyError("missing commitment C%s-0" % wname[1:])
        return out

    def lc_of(token):
        if token[0] == "W":
            return commitments["C%s-0" % token[1:]]
        return be_to_scalar(instance[token])

    def derived(index, sub, upto):
        out = []
        for k in range(upto):
            key = "D%d-%d-%d" % (index, sub, k)
            if key not in commitments:
                break
            out.append(commitments[key])
        return out

    def hash_witness(wn, index, sub, cs):                                 # verifier.rs:426-444 -> image Variable
        d = derived(index, sub, 3)
        MimcHash256(d[0]).verify(cs, all_commitments(wn), d[1:])
        return d[0]

    with open(name + ".gadgets") as f:
        lines = [l.rstrip("\n") for l in f]

    def do_gadget(line, index, cs):
        parts = line.split()
        op = parts[0]
        if op == "BOUND":                                                 # verifier.rs:188-205
            BoundsCheck(instance[parts[2]], instance[parts[3]]).verify(cs, [commitments["C%s-0" % parts[1][1:]]], derived(index, 0, 2))
        elif op == "HASH":                                                # verifier.rs:207-230
            MimcHash256(lc_of(parts[1])).verify(cs, all_commitments(parts[2]), derived(index, 0, 2))
        elif op == "MERKLE":                                              # verifier.rs:232-260
            root = lc_of(parts[1])
            inst_names, wit_names, pattern = parse_tree(line.split(None, 2)[2])
            inst_lcs = [mimc_hash(instance[i]) for i in inst_names]
            wit_lcs = [hash_witness(wn, index, sub, cs) for sub, wn in enumerate(wit_names)]
            MerkleTree256(root, inst_lcs, wit_lcs, pattern).verify(cs, [], [])
        elif op == "EQUALS":                                              # verifier.rs:262-280
            left, right = (parts[1], parts[2]) if parts[1][0] == "W" else (parts[2], parts[1])
            right_lcs = all_commitments(right) if right[0] == "W" else be_to_scalars(instance[right])
            Equality(right_lcs).verify(cs, all_commitments(left), [])
        elif op == "LESS_THAN":                                           # verifier.rs:282-301
            LessThan(commitments["C%s-0" % parts[1][1:]], None, commitments["C%s-0" % parts[2][1:]], None).verify(cs, [], derived(index, 0, 2))
        elif op == "UNEQUAL":                                             # verifier.rs:303-332
            left, right = (parts[1], parts[2]) if parts[1][0] == "W" else (parts[2], parts[1])
            lv = all_commitments(left)
            right_lcs = all_commitments(right) if right[0] == "W" else be_to_scalars(instance[right])
            d = [commitments["D%d-0-%d" % (index, k)] for k in range(2 * len(lv) + 1)]
            Inequality(right_lcs, None).verify(cs, lv, d)
        elif op == "SET_MEMBER":                                          # verifier.rs:334-424
            member, elems = parts[1], parts[2:]
            m_lcs = all_commitments(member) if member[0] == "W" else be_to_scalars(instance[member])
            m_lc = m_lcs[0]
            hashing = False
            w_vars, i_lcs = [], []
            for e in elems:
                if e[0] == "W":
                    cw = all_commitments(e)
                    if len(cw) == 1:
                        w_vars.append(cw[0])
                    else:
                        hashing = True
                else:
                    sc_ = be_to_scalars(instance[e])
                    if len(sc_) == 1:
                        i_lcs.append(sc_[0])
                    else:
                        hashing = True
            if len(m_lcs) > 1:
                hashing = True
            d = [commitments["D%d-0-%d" % (index, k)] for k in range(len(elems))]
            if hashing:
                sub = 1
                if member[0] == "W":
                    m_lc = hash_witness(member, index, sub, cs); sub += 1
                else:
                    m_lc = mimc_hash(instance[member])
                w_vars, i_lcs = [], []
                for e in elems:
                    if e[0] == "W":
                        w_vars.append(hash_witness(e, index, sub, cs)); sub += 1
                    else:
                        i_lcs.append(mimc_hash(instance[e]))
            SetMembership(m_lc, None, i_lcs, None).verify(cs, w_vars, d)
        else:
            raise ValueError("unknown gadget line: %r" % line)

    def run_block(i, cs, closing, depth=0):
        if depth > MAX_OR_NESTING:
            raise ValueError("OR blocks nested deeper than %d" % MAX_OR_NESTING)
        while i < len(lines):
            line, index = lines[i], i
            i += 1
            parts = line.split()
            if not parts:
                continue
            op = parts[0]
            if closing is not None and op == closing:
                return i
            if op == "}":
                cs.rewind()
            elif op == "OR":                                              # verifier.rs:162-186
                child = ConstraintBuffer(cs, False)
                i = run_block(i, child, "]", depth + 1)
                or_conjunction(cs, child)
            elif op in ("[", "{"):
                pass
            else:
                do_gadget(line, index, cs)
        if closing is not None:
            raise ValueError("unexpected end of input")
        return i
    run_block(0, v, None)
    return v, transcript


def verifier(name, ctx=None, flags=0, quiet=False):
    ctx = ctx or Context(0)
    v, _ = assemble_verifier(name)
    with open(name + ".proof", "rb") as f:
        proof = f.read()
    ok = v.is_valid(proof, ctx, round_pow2(v.get_num_vars()), os.urandom(32), flags)
    if not quiet:
        print("true" if ok else "false")                                  # verifier.rs:91-100
    return ok


def read_batch(path):
    """stems of a batch file: one per line, blank lines and # comments skipped"""
    with open(path) as f:
        return [l.strip() for l in f if l.strip() and not l.lstrip().startswith("#")]


def spawn_batch_ranks(mode, batch_file, gpus):
    """`--gpus N` without a torchrun environment: start N ranks of this module (fresh processes, before anything touches the GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "bulletproofs_gadgets_amd.cli", mode, "--batch", batch_file, "--gpus", str(gpus)]
    # the ranks' own verdict (0, 1 = a proof rejected, 101 = a stem failed) travels through a file: the elastic launcher turns every non-zero exit of a
    # rank into 1 and a page of its own diagnostics, so the ranks of a batch that RAN exit 0 and rank 0 leaves the batch's exit code here
    import tempfile
    fd, rc_file = tempfile.mkstemp(prefix="bpg_batch_rc_")
    os.close(fd)
    env["BPG_BATCH_RC_FILE"] = rc_file
    try:
        launcher = subprocess.run(cmd, env=env).returncode
        text = open(rc_file).read().strip()
        return int(text) if launcher == 0 and text else (launcher or 101)
    finally:
        os.unlink(rc_file)


def run_batch(mode, batch_file, gpus=1, seed=None, rng_seed=None, out=sys.stdout):
    """One rank of a batch (the whole batch when gpus == 1).  Returns the list of per-stem results in file order on every rank:
    prover: (stem, constraints, proof bytes); verifier: (stem, bool)."""
    stems = read_batch(batch_file)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit("cli --batch: --gpus %d but WORLD_SIZE=%d (one rank per GPU; without torchrun the driver spawns them itself)" % (gpus, world))
    dist, coll_device, device = None, "cpu", 0
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("BPG_BATCH_BACKEND", "nccl")              # gloo: rehearsal on fewer GPUs than ranks (the ranks then share devices)
        ndev = torch.cuda.device_count()
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend == "nccl":
            if world > ndev:
                raise SystemExit("cli --batch: %d ranks but %d GPUs (BPG_BATCH_BACKEND=gloo rehearses on fewer)" % (world, ndev))
            device, coll_device = local_rank, "cuda"
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            device = local_rank % max(ndev, 1)
            dist.init_process_group(backend)
    from .batch import gather_proofs, shard_indices
    ctx = Context(device)
    mine = shard_indices(len(stems), rank, world)
    local = {}
    FAILED = (1 << 64) - 1                                                # constraint count of a stem that could not be proved
    for i in mine:
        # ONE stem per try (as in csrc/cli_main.cpp): a missing or garbled file ends that stem only - it would end only its own run in the reference's
        # CI (one prover process per stem, .github/workflows/integration_tests.yml:19-58) - and, with several ranks, must never keep this rank
        # away from the collective the others are waiting in
        try:
            if mode == "prover":
                p, proof = prover(stems[i], ctx, seed=seed, rng_seed=rng_seed, quiet=True)
                local[i] = (p.num_constraints(), proof)
            else:
                local[i] = 1 if verifier(stems[i], ctx, quiet=True) else 0
        except Exception as e:      # noqa: BLE001 - reported per stem below
            sys.stderr.write("%s --batch (rank %d): %s: %s\n" % (mode, rank, stems[i], e))
            local[i] = (FAILED, b"") if mode == "prover" else 2
    if mode == "prover":
        # the finished proofs cross the fabric as fixed-size records: constraints (8 B) | length (4 B) | proof bytes, padded to the longest proof of the batch
        longest = max([len(pr) for _, pr in local.values()] or [0])
        if dist is not None:
            import torch
            t = torch.tensor([longest], dtype=torch.int64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            longest = int(t.item())
        rec_len = 12 + longest
        recs = {i: q.to_bytes(8, "little") + len(pr).to_bytes(4, "little") + pr + bytes(longest - len(pr)) for i, (q, pr) in local.items()}
        allrecs = gather_proofs(recs, len(stems), rec_len, dist, device=coll_device)
        results = []
        for stem, r in zip(stems, allrecs):
            n = int.from_bytes(r[8:12], "little")
            results.append((stem, int.from_bytes(r[:8], "little"), r[12:12 + n]))
        results = [(stem, None, None) if q == FAILED else (stem, q, pr) for stem, q, pr in results]
        if rank == 0:
            for stem, q, pr in results:
                if q is None:
                    out.write("%s: FAILED\n" % stem)
                else:
                    out.write("%s: %d constraints, %d-byte proof, sha256 %s\n" % (stem, q, len(pr), hashlib.sha256(pr).hexdigest()[:16]))
    else:
        recs = {i: bytes([v]) for i, v in local.items()}
        allrecs = gather_proofs(recs, len(stems), 1, dist, device=coll_device)
        results = [(stem, None if r == b"\x02" else r == b"\x01") for stem, r in zip(stems, allrecs)]
        if rank == 0:
            for stem, ok in results:
                out.write("%s: %s\n" % (stem, "FAILED" if ok is None else ("true" if ok else "false")))
    out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return results


def main(argv=None):
    argv = argv if argv is not None else sys.argv[1:]
    if "--batch" in argv:
        gpus = 1
        ok = len(argv) in (3, 5) and argv[0] in ("prover", "verifier") and argv[1] == "--batch"
        if ok and len(argv) == 5:
            ok = argv[3] == "--gpus" and argv[4].isdigit() and int(argv[4]) >= 1
            gpus = int(argv[4]) if ok else 1
        if not ok:
            print(__doc__)
            return 2
        if gpus > 1 and "WORLD_SIZE" not in os.environ:
            return spawn_batch_ranks(argv[0], argv[2], gpus)
        # tests pin the randomness the way the native driver does (csrc/cli_main.cpp): BPG_CLI_SEED = blinding stream, BPG_CLI_RNG_SEED = hex rng seed
        seed = os.environ["BPG_CLI_SEED"].encode() if "BPG_CLI_SEED" in os.environ else None
        rng_seed = bytes.fromhex(os.environ["BPG_CLI_RNG_SEED"]).ljust(32, b"\0")[:32] if "BPG_CLI_RNG_SEED" in os.environ else None
        res = run_batch(argv[0], argv[2], gpus, seed=seed, rng_seed=rng_seed)
        if any(r[1] is None for r in res):
            rc = 101                                                      # a stem could not be processed: the exit code of the reference's panic
        else:
            rc = 0 if argv[0] == "prover" or all(ok for _, ok in res) else 1
        rc_file = os.environ.get("BPG_BATCH_RC_FILE")                     # a rank started by spawn_batch_ranks: see there
        if rc_file:
            if int(os.environ.get("RANK", "0")) == 0:
                with open(rc_file, "w") as f:
                    f.write(str(rc))
            return 0
        return rc
    if len(argv) != 2 or argv[0] not in ("prover", "verifier"):
        print(__doc__)
        return 2
    # a run makes ONE proof per stem: Context()'s default, the one-shot profile (csrc/cli_main.cpp does the same)
    if argv[0] == "prover":
        prover(argv[1])
        return 0
    return 0 if verifier(argv[1]) else 1


if __name__ == "__main__":
    sys.exit(main())
