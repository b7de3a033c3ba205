"""pytest configuration: `gpu` marker, repo root on sys.path, shared fixtures."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    return json.loads((GOLDEN / "kat_counts.json").read_text())


@pytest.fixture(scope="session")
def small_n():
    return dict(np.load(GOLDEN / "small_n.npz"))


@pytest.fixture(scope="session")
def panels_golden():
    return dict(np.load(GOLDEN / "panels.npz"))


@pytest.fixture(scope="session")
def drivers():
    return json.loads((GOLDEN / "drivers.json").read_text())


# the panels of tests/golden/make_golden.py: name -> (n_snps, n_hap, seed, miss)
PANELS = {
    "c1_64x5008": (64, 5008, 7, 0.0),
    "miss_32x5008": (32, 5008, 1, 0.01),
    "eur_64x1008": (64, 1008, 7, 0.0),
    "tie_96x100": (96, 100, 1, 0.0),
    "odd_96x37": (96, 37, 7, 0.02),
}


@pytest.fixture(scope="session")
def panel_codes():
    from ld_tools_amd import synth

    return {name: synth.synth_codes_host(n, h, seed=seed, miss=miss) for name, (n, h, seed, miss) in PANELS.items()}


def tri_pairs(n):
    """(rows, cols) of the strict lower triangle in the fixtures' order: for i: for j < i."""
    rows = np.concatenate([np.full(i, i, dtype=np.int64) for i in range(n)]) if n > 1 else np.zeros(0, np.int64)
    cols = np.concatenate([np.arange(i, dtype=np.int64) for i in range(n)]) if n > 1 else np.zeros(0, np.int64)
    return rows, cols


def realise(n, n11, a1, r1, a2, r2):
    """Two code lists (codes 0, 1, 2) with exactly these counts (as tests/golden/make_golden.py builds them)."""
    o1, o2 = n - a1 - r1, n - a2 - r2
    assert min(o1, o2, n11, a1 - n11, a2 - n11) >= 0
    row = {1: a1 - n11, 0: r1, 2: o1}
    col = {1: a2 - n11, 0: r2, 2: o2}
    cells = {(1, 1): n11}
    for (rc, cc) in [(1, 0), (1, 2), (0, 1), (2, 1), (0, 0), (0, 2), (2, 0), (2, 2)]:
        take = min(row[rc], col[cc])
        cells[(rc, cc)] = take
        row[rc] -= take
        col[cc] -= take
    assert not any(row.values()) and not any(col.values())
    g1, g2 = [], []
    for (rc, cc), k in cells.items():
        g1 += [rc] * k
        g2 += [cc] * k
    return g1, g2
