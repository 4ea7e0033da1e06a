"""pytest configuration: `gpu` marker + shared fixture loader."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_names(prefix: str = ""):
    return sorted(p.stem for p in GOLDEN.glob("*.npz") if p.stem.startswith(prefix))


def load_golden(name: str):
    z = np.load(GOLDEN / f"{name}.npz")
    return {
        "name": name,
        "queries": z["queries"],
        "subjects": z["subjects"],
        "scores": z["scores"],
        "variant": str(z["variant"]),
        "k": int(z["k"]),
    }


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O
