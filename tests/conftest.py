import sys, pathlib
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))


def pytest_addoption(parser):
    parser.addoption("--run-sanitizers", action="store_true", default=False,
                     help="run the long sanitizer legs of tests/test_sanitizers.py (the whole host side under ASan/UBSan and TSan, the 10^5-case fuzz); "
                          "`make -C tests/hostcheck sanitize` builds what they need and passes this flag")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    return json.loads((ROOT / "tests" / "golden" / "primitives.json").read_text())
