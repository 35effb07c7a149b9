"""Build libbpg_hip.so (HIP kernels + C ABI + C++ host mirror) for gfx950, in-tree."""
import os
import pathlib
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libbpg_hip.so"
SOURCES = [CSRC / "engine.hip", CSRC / "capi.hip"]
# native file drivers (reference src/bin/prover.rs, src/bin/verifier.rs) over the C ABI: one executable, two names
CLI_SRC = CSRC / "cli_main.cpp"
CLI_BINS = [PKG / "bin" / "bpg_prover", PKG / "bin" / "bpg_verifier"]


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


def build_cli(verbose=False):
    """g++ csrc/cli_main.cpp against include/bpg.h + libbpg_hip.so (rpath = the package directory)."""
    if all(b.exists() and b.stat().st_mtime >= max(CLI_SRC.stat().st_mtime, LIB.stat().st_mtime) for b in CLI_BINS):
        return CLI_BINS
    CLI_BINS[0].parent.mkdir(exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-g", "-rdynamic", "-std=c++17", "-pthread", str(CLI_SRC), "-o", str(CLI_BINS[0]), "-L", str(PKG), "-lbpg_hip",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    import shutil
    shutil.copy2(CLI_BINS[0], CLI_BINS[1])
    return CLI_BINS


def build(force=False, verbose=False):
    if not force and not needs_build():
        build_cli(verbose)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
           "-Xarch_host", "-march=x86-64-v3",     # host side only: BMI2 rotates / ANDN for the Keccak chain
           "-o", str(LIB)] + [str(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=str(CSRC))
    build_cli(verbose)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
