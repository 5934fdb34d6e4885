"""N>1 path on CPU: world_size 2 and 3 over gloo (tests/partition_worker.py)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world,case,port", [(2, "grid", 29731), (3, "grid", 29732), (2, "ragged", 29733)])
def test_sharded_jacobi_over_gloo(product, world, case, port):
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "partition_worker.py"), str(r), str(world), str(port), case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_strips_are_row_blocks_cut_at_chunk_boundaries(product):
    """512-wide grid, 8 ranks: every rank owns 64 full rows of every layer; halos are one row."""
    m = cm.catchment_model(128, 64, 4)
    cm.build(product, m, finalize=False)
    assert 250 * m.n < int(product.lib.sf3d_host_bytes()) < 600 * m.n          # the resident staging model: ~330 B per node, counted page by page
    owner = product.owner_map(4, m.n).reshape(4, 64, 128)
    for r in range(4):
        assert (owner[:, 16 * r:16 * (r + 1), :] == r).all()           # 64 rows / 4 ranks, all layers (columns stay whole)
    up = product.halo_list(1, 4, 0, 1)                                   # rank 1 receives from rank 0: row 15 of each layer
    assert len(up) == 128 * 4 and (np.sort(up) == up).all()
    assert len(product.halo_list(1, 4, 3, 1)) == 0                       # strips only talk to adjacent strips
    assert np.array_equal(product.halo_list(0, 4, 1, 0), up)             # what 0 sends to 1 is what 1 receives from 0
    assert product.lib.sf3d_dist_prepare(5, 4) == capi.PARAMETER_ERROR
    product.lib.sf3d_clean()
