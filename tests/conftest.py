import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vk():
    """The product package (directory name has hyphens, so import by string)."""
    return importlib.import_module("vickers-hardness-unet_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import unet_oracle
    return unet_oracle


def golden_loops_batches():
    """The seeded (x, y, names) loaders of tests/golden/loops_ref.json (defined once, next to the script that ran the reference's
    own loops over them; importing that script reads nothing from /root/reference)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vk_make_golden", GOLDEN / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.loops_batches()
