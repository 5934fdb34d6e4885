"""Run one scenario on the wrapped reference and on the oracle in THIS (fresh) process and exit 0
when every array of the two traces is bitwise equal.  A fresh process is needed because the
reference keeps its solver parameters across re-initialisations."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from criteria3d_amd import capi  # noqa: E402
from tests import checkers
from tests.scenarios import run_scenario  # noqa: E402


def main(name):
    try:
        ref = checkers.load_reference()
    except Exception as e:  # noqa: BLE001
        print(f"SKIP {e}")
        return 77
    a = run_scenario(checkers.load_oracle(), name, threads=1)
    b = run_scenario(ref, name, threads=1)
    def same(k):
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if k.startswith("flux_h"):          # stale-matrix-slot deviation of the WaterLiquidIsothermal entry, see test_oracle_golden.py
            x = x.copy(); stale = (x[..., 5] == 0.0) & (y[..., 5] != 0.0); x[..., 5][stale] = y[..., 5][stale]
        return np.array_equal(x, y)
    bad = [k for k in b if not same(k)]
    print("DIFF " + ",".join(bad) if bad else "EQUAL")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
