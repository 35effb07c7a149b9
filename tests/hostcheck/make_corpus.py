#!/usr/bin/env python3
"""Seed corpus of tests/hostcheck/fuzz_cli: one file per stem of tests/golden/resources (the reference's own test inputs) in the target's
five-section form (gadgets, inst, wtns, coms, proof separated by "===="), with plausible .coms / .proof sections so that the verifier side
gets past its first reads; plus a few hand-made edge cases (empty files, unknown gadget, unbalanced OR block, odd hex, long values, a tree of twenty thousand opening brackets)."""
import hashlib, pathlib, sys
out = pathlib.Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
res = pathlib.Path(__file__).resolve().parent.parent / "golden" / "resources"
SEP = "\n====\n"
h32 = lambda *a: hashlib.sha256(repr(a).encode()).hexdigest()
stems = sorted({p.stem for p in res.glob("*.gadgets")})
for s in stems:
    g = (res / (s + ".gadgets")).read_text(); i = (res / (s + ".inst")).read_text() if (res / (s + ".inst")).exists() else ""
    w = (res / (s + ".wtns")).read_text() if (res / (s + ".wtns")).exists() else ""
    # commitments the verifier will look up: C<k>-<j> for every witness scalar, D<line>-<sub>-<k> for derived values (any 32 bytes parse)
    coms = []
    for line in w.splitlines():
        if " = 0x" not in line:
            continue
        name, val = line.split(" = 0x")
        for j in range(max(1, (len(val) // 2 + 30) // 31)):
            coms.append("C%s-%d = 0x%s" % (name[1:], j, h32(name, j)))
    for ln in range(len(g.splitlines())):
        for sub in range(3):
            for k in range(4):
                coms.append("D%d-%d-%d = 0x%s" % (ln, sub, k, h32(ln, sub, k)))
    (out / (s + ".seed")).write_text(g + SEP + i + SEP + w + SEP + "\n".join(coms) + "\n" + SEP + "\x00" * 1024)
edge = {
    "empty": SEP * 4,
    "unknown_gadget": "FROBNICATE W0 I0\n" + SEP + "I0 = 0x01\n" + SEP + "W0 = 0x02\n" + SEP + "C0-0 = 0x" + "11" * 32 + "\n" + SEP,
    "open_or": "OR [\n{\nBOUND W0 I0 I1\n" + SEP + "I0 = 0x00\nI1 = 0xff\n" + SEP + "W0 = 0x05\n" + SEP + SEP,
    "odd_hex": "BOUND W0 I0 I1\n" + SEP + "I0 = 0x0\nI1 = 0xfff\n" + SEP + "W0 = 0x5\n" + SEP + SEP,
    "long_bound": "BOUND W0 I0 I1\n" + SEP + "I0 = 0x00\nI1 = 0x" + "ff" * 40 + "\n" + SEP + "W0 = 0x" + "05" * 70 + "\n" + SEP + SEP,
    "tree_deep": "MERKLE I0 " + "(" * 20000 + "W0" + "\n" + SEP + "I0 = 0x01\n" + SEP + "W0 = 0x03\n" + SEP + SEP,
    "tree_garbage": "MERKLE I0 ((W0 I1) (W1\n" + SEP + "I0 = 0x01\nI1 = 0x02\n" + SEP + "W0 = 0x03\nW1 = 0x04\n" + SEP + SEP,
    "set_member": "SET_MEMBER W0 I0 W1 I1\n" + SEP + "I0 = 0x" + "aa" * 40 + "\nI1 = 0x07\n" + SEP + "W0 = 0x07\nW1 = 0x" + "bb" * 33 + "\n" + SEP + SEP,
}
for k, v in edge.items():
    (out / (k + ".seed")).write_text(v)
print("corpus: %d seeds in %s" % (len(stems) + len(edge), out))
