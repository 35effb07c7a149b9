"""Build libbpg_hip.so (HIP kernels + C ABI + C++ host mirror) for gfx950, in-tree."""
import os
import pathlib
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libbpg_hip.so"
SOURCES = [CSRC / "engine.hip", CSRC / "capi.hip"]


def _deps():
    out = list(SOURCES)
    for pat in ("*.hpp", "hip/*.cuh", "host/*.hpp", "host/*.inc"):
        out += list(CSRC.glob(pat))
    out.append(PKG.parent / "include" / "bpg.h")
    return out


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
           "-Xarch_host", "-march=x86-64-v3",     # host side only: BMI2 rotates / ANDN for the Keccak chain
           "-o", str(LIB)] + [str(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
