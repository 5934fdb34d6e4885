"""Live cross-check of the oracle against the unmodified reference, when oracle/_ref is
loadable (in the build container; skipped on machines without the prebuilt library or Qt).
Each case runs in a fresh interpreter (tests/live_compare.py) - the reference cannot be reset."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.scenarios import run_scenario

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("name", ["c1_column", "ragged_edge_cases", "ragged_arithmetic_vg", "ragged_geometric", "het_patches",
                                  "heat_column_latent", "heat_column_conduction"])
def test_bitwise_equal(name):
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "live_compare.py"), name], capture_output=True, text=True)
    if p.returncode == 77:
        pytest.skip(p.stdout.strip())
    assert p.returncode == 0, p.stdout + p.stderr


def test_thread_count_does_not_change_the_oracle(oracle):
    """The restatement sums in index order whatever the thread count (deterministic reductions)."""
    a, b = run_scenario(oracle, "ragged_edge_cases", threads=1), run_scenario(oracle, "ragged_edge_cases", threads=4)
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
