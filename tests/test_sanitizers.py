"""Sanitizer runs of the product's HOST side (SURVEY.md section 5 "race detection / sanitizers"; the reference is safe Rust, src/lib.rs:1, so
memory and thread safety are what a C++ replacement has to show).  Build container only - there are no GPU sanitizers on this pool.

Always (part of the CPU suite, ~40 s): the chain threads and their lock-free publication protocol, Merlin and the lockstep Keccak
(csrc/host/chain.hpp, merlin.hpp) under -fsanitize=thread and -fsanitize=address,undefined through tests/hostcheck/san_chain.cpp.

With --run-sanitizers (or BPG_RUN_SANITIZERS=1; `make -C tests/hostcheck sanitize` builds the artefacts and passes the flag):
the device-less pytest files against libbpg_hip_asan.so and libbpg_hip_tsan.so (the WHOLE library, host side instrumented), the native file
drivers' bad-input handling under ASan, and the 10^5-case libFuzzer run over the .gadgets/.inst/.wtns/.coms/.proof readers."""
import os
import pathlib
import subprocess
import sys
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
HC = ROOT / "tests" / "hostcheck"
OUT = HC / "build"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _long(request):
    if not (request.config.getoption("--run-sanitizers") or os.environ.get("BPG_RUN_SANITIZERS")):
        pytest.skip("long sanitizer leg: run `make -C tests/hostcheck sanitize` (or pytest --run-sanitizers)")


def _rt(name):
    return subprocess.check_output([CLANG, "-print-file-name=libclang_rt.%s-x86_64.so" % name], text=True).strip()


def test_chain_threads_under_tsan_and_asan():
    """Five chain threads (single-lane and lockstep) serve twelve blinding streams of several 'contexts' while twelve consumer threads adopt them
    the way Engine::prove does (block by block through uploaded_blocks / err, then the generator snapshot through produced), one stream with an
    injected upload failure, some consumers stopping early: no data race, no memory error, and every byte equals a single-threaded redraw."""
    subprocess.check_call(["make", "-C", str(HC), str(OUT / "san_chain.tsan"), str(OUT / "san_chain.asan")], stdout=subprocess.DEVNULL)
    r = subprocess.run([str(OUT / "san_chain.tsan"), "2", "12", "30000"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
    assert r.returncode == 0 and "san_chain ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
    r = subprocess.run([str(OUT / "san_chain.asan"), "1", "12", "30000"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "san_chain ok" in r.stdout and "Sanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]


DEVICELESS = ["tests/test_host_logic.py", "tests/test_assembly_fixtures.py", "tests/test_cli_batch.py", "tests/test_oracle_primitives.py"]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_deviceless_suite_against_the_sanitized_library(kind, request):
    """The device-less pytest files (host scalar / Merlin / R1CS assembly of every gadget and OR block / C-ABI argument checking / chain-pool API /
    knob validation / batch drivers without a GPU) with libbpg_hip.so replaced by a build of the same sources whose host side is compiled with
    -fsanitize=address,undefined resp. -fsanitize=thread (hipcc -Xarch_host): any report fails the run."""
    _long(request)
    lib = OUT / ("libbpg_hip_%s.so" % kind)
    if not lib.exists():
        subprocess.check_call(["make", "-C", str(HC), str(lib)])
    env = dict(os.environ, BPG_LIB_PATH=str(lib), LD_PRELOAD=_rt(kind), PYTHONMALLOC="malloc")
    if kind == "asan":
        # leaks: CPython itself never frees everything; link order: the interpreter is not instrumented, the runtime comes through LD_PRELOAD
        env.update(ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_odr_violation=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    else:
        env.update(TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1:report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + DEVICELESS, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=3000)
    tail = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, tail
    assert "Sanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail


def test_fuzz_of_the_file_readers(request):
    """10^5 libFuzzer executions (ASan + UBSan, coverage-guided, seeded with the reference's own test files) of the native drivers' readers and the
    gadget assembly behind them on a device-less prover / verifier: no crash, no sanitizer report, no timeout."""
    _long(request)
    r = subprocess.run(["make", "-C", str(HC), "run-fuzz"], capture_output=True, text=True, timeout=3400)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    done = sum(int(l.split()[-1]) for f in OUT.glob("fuzz-*.log") for l in f.read_text(errors="replace").splitlines() if "stat::number_of_executed_units" in l)
    assert done >= 90000, done
