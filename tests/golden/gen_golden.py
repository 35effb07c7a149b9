#!/usr/bin/env python3
"""Generate tests/golden/primitives.json from pyref.py (pure Python big-int arithmetic + hashlib).

Before writing anything the script pins pyref itself against values that do NOT come from this
repository:
  * RFC 9496 A.1 (generator multiples 1B, 2B) and the dalek Pedersen blinding base
  * the Merlin crate's published "equivalence_simple" transcript vector
  * the reference's own known-answer values: src/mimc_hash/mimc.rs:104-143,
    src/mimc_hash/mimc_hash_gadget.rs:236-251, tests/resources/merkle_tree.inst:1,3,5 with
    merkle_tree.wtns:1-2, and the 512-leaf root + level digests in
    src/merkle_tree/merkle_tree_gadget.rs:476-503 (leaf W1 at :126-131).
Run:  python3 tests/golden/gen_golden.py      (a few seconds)
"""
import json, pathlib, hashlib
import pyref as R

here = pathlib.Path(__file__).resolve().parent
consts = [int.from_bytes(bytes.fromhex(l), "little") for l in (here / "mimc_rc769.hex").read_text().split()]
hx = lambda b: b.hex()
sc = lambda x: (x % R.L).to_bytes(32, "little").hex()

# ---- pins from outside this repo
B, Bb = R.pedersen_gens()
assert hx(B.compress()) == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
assert hx((2 * B).compress()) == "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"
assert hx(Bb.compress()) == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"
assert hx((0 * B).compress()) == "00" * 32
t = R.Transcript(b"test protocol")
t.append_message(b"some label", b"some data")
assert hx(t.challenge_bytes(b"challenge", 32)) == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"

# ---- reference KATs
be = lambda x: (x % R.L).to_bytes(32, "big").hex()
assert be(R.mimc_hash(bytes.fromhex("38535450433043546f313877615a6a423663"), consts)) == \
    "0d2203069ac15f58172bae1b3af98d8982deef9df37482c1a920b8832ee813a4"          # mimc.rs:105-121
assert be(R.mimc_hash(b"The quick brown fox jumps over t", consts)) == \
    "01245409f28ae2f076077d4a40bd91551b3a03b1ad8adb2b1da116d29c60a85c"          # mimc.rs:124-142
john = R.mimc_hash(b"John", consts)
doe = R.mimc_hash(b"Doe", consts)
n1 = R.mimc_sponge([john, john], consts)   # MERKLE I0 (W0 I1): both leaves "John" (merkle_tree.inst:1-2, .wtns:1)
n2 = R.mimc_sponge([doe, doe], consts)
assert be(n1) == "0e65ad60f91829a76f08c39e4eec78c82dd0686c733ec5afc25ca28ae4628898"
assert be(n2) == "0cdc849ef63dd4e0d3a984a1c8f3bcbe4ae59c378b8d1726181433511a37e8b9"
assert be(R.mimc_sponge([n1, n2], consts)) == "0b33a0e69996bf60542d94951136e4246b15591e3e47d7aeb1a7822ee96101c8"
W1 = bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")
levels = ["0b79280bd08952b2f43c000fa7ee45d0f73c0242a34033e9fde3cac80deaff7c",
          "0f06bee0afba3bfe751787721eafd769e993e1700cde9b7b2146fc508efc54e5",
          "04af68c673b12851f92603154c51a9ea1714a855686f275b54539a8696d6ce60",
          "004ce529f3e16d7c7d40fd72033ccdb351b710d0aab96ab350fb206202a0328b",
          "0fe33807557b26124c6f60abede601a60129879441c08d8ea940cf45086e1cce",
          "0a3bcac677f47d10383e7efd397d0f71b951704504b7a9ad81848fdc29855f3a",
          "049057939c976063cfaed9e15fc02c8dbd997e12f9b919a97781870ad689bd41",
          "0c61fcdd0add4eb6d44de2be6ef2353871696ed586af8aa5fd1b54478c989fe1",
          "038c137beec8e2edfb5c48cbd063f04e569139d2221a4eb7befb85aa1bf8ba40"]
h = R.scalars_be(W1)[0]
for want in levels:
    h = R.mimc_sponge([h, h], consts)
    assert be(h) == want, want

# ---- fixtures
out = {"_provenance": "tests/golden/gen_golden.py (pure Python; see its docstring for the external pins)"}
out["pedersen"] = {"B": hx(B.compress()), "B_blinding": hx(Bb.compress())}
G, H = R.bp_gens(8)
out["bp_gens"] = {"G": [hx(p.compress()) for p in G], "H": [hx(p.compress()) for p in H]}
out["ristretto_multiples"] = [hx((k * B).compress()) for k in range(0, 17)]
# one-way map vectors: label -> sha512 -> point (RFC 9496 A.3 style inputs, hashed with SHA-512)
owm = []
for lab in [b"Ristretto is traditionally a short shot of espresso coffee", b"made with the normal amount of ground coffee but extracted with",
            b"about half the amount of water in the same amount of time", b"by using a finer grind."]:
    u = hashlib.sha512(lab).digest()
    owm.append({"uniform": hx(u), "point": hx(R.from_uniform_bytes(u).compress())})
assert owm[0]["point"] == "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"   # RFC 9496 A.3 vector 1
out["one_way_map"] = owm
# pedersen commitments
ped = []
for v, r in [(0x43, 1), (0x43, 2**252 - 1), (0, 0), (R.L - 1, R.L - 2), (2**254 + 12345, 7)]:
    ped.append({"v": v.to_bytes(32, "little").hex(), "r": (r % R.L).to_bytes(32, "little").hex(),
                "commit": hx((v * B + r * Bb).compress())})
assert ped[0]["commit"] == "e8efa9211c294cb7a6991b7406b329141c17f568446ef6ee8f85a984590b4827"
out["pedersen_commit"] = ped
# small MSM: sum s_i G_i + t_i H_i
ss = [int.from_bytes(hashlib.sha512(b"msm-s%d" % i).digest(), "little") % R.L for i in range(8)]
tt = [int.from_bytes(hashlib.sha512(b"msm-t%d" % i).digest(), "little") % R.L for i in range(8)]
acc = R.Point.identity()
for i in range(8):
    acc = acc + ss[i] * G[i] + tt[i] * H[i]
out["msm8"] = {"s": [sc(x) for x in ss], "t": [sc(x) for x in tt], "result": hx(acc.compress())}
# transcript chain used by the prover (labels as in dalek bulletproofs transcript.rs)
t = R.Transcript(b"example")
t.append_message(b"dom-sep", b"r1cs v1")
V = bytes.fromhex(ped[0]["commit"])
t.append_message(b"V", V)
t.append_u64(b"m", 1)
st_after_m = t.strobe.state_bytes()
rng = t.build_rng([(b"v_blinding", (1).to_bytes(32, "little"))], bytes(range(32)))
draws = [sc(rng.random_scalar()) for _ in range(5)]
y = t.challenge_scalar(b"y")
assert sc(y) == "a7b59b392d5f793b6cdee2693182a312ef0af6e58d429637c705880dba8aed0d"
out["transcript_chain"] = {"label": "example", "V": hx(V), "state_after_m": hx(st_after_m),
                           "rng_seed": bytes(range(32)).hex(), "rng_scalars": draws, "y": sc(y),
                           "z": sc(t.challenge_scalar(b"z"))}
# wide reduction + scalar arithmetic vectors
wide = []
for i in range(6):
    b = hashlib.sha512(b"wide%d" % i).digest() if i else b"\xff" * 64
    wide.append({"in": hx(b), "out": sc(int.from_bytes(b, "little"))})
out["sc_wide"] = wide
a = int.from_bytes(hashlib.sha256(b"a").digest(), "little") & (2**255 - 1)
b = int.from_bytes(hashlib.sha256(b"b").digest(), "little") & (2**255 - 1)
out["sc_arith"] = {"a": a.to_bytes(32, "little").hex(), "b": b.to_bytes(32, "little").hex(), "mul": sc(a * b),
                   "add": sc(a + b), "sub": sc(a - b), "inv_a": sc(pow(a % R.L, R.L - 2, R.L))}
# MiMC
out["mimc"] = {"kat1_in": "38535450433043546f313877615a6a423663",
               "kat1_be": "0d2203069ac15f58172bae1b3af98d8982deef9df37482c1a920b8832ee813a4",
               "kat2_in": b"The quick brown fox jumps over t".hex(),
               "kat2_be": "01245409f28ae2f076077d4a40bd91551b3a03b1ad8adb2b1da116d29c60a85c",
               "kat3_in": "43", "kat3_be": be(R.mimc_hash(b"\x43", consts)),
               "john_be": be(john), "doe_be": be(doe), "node_john": be(n1), "node_doe": be(n2),
               "leaf512_be": W1.hex(), "levels512_be": levels}
assert out["mimc"]["kat3_be"] == "0cfb0c17618211c607febf703ac3f3078f7d96798fae9d4a1682bc592f7cb126"  # example.wtns:3 / combine_gadgets.rs:34-39
(here / "primitives.json").write_text(json.dumps(out, indent=1) + "\n")
print("wrote primitives.json")
