"""Registers, scratch and LDS of the hot kernels, read from the code object inside the built product library (no GPU needed).

The host side PLANS with these numbers - the patch-height cost model of the paired sweep assumes floor(4 * 6 / (W + 1)) blocks of W + 1
waves per CU (sf3d_host_build.inc, "perCU"), the resident grids of k_props / k_assemble / k_post assume 5 / 4 / 8 waves per SIMD
 - and a change that makes a kernel spill or lose a resident block shows up as a slower bench line long after the edit.
This test makes it show up at once."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

from criteria3d_amd import build

LLVM = Path("/opt/rocm/lib/llvm/bin")
LDS_PER_CU = 160 * 1024          # MI355X_MICROARCH.md
VGPRS_PER_SIMD_LANE = 512


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not (LLVM / "llvm-objdump").exists() or not (LLVM / "llvm-readelf").exists():
        pytest.skip("no llvm-objdump / llvm-readelf in this image")
    lib = build.build_product()
    d = tmp_path_factory.mktemp("codeobj")
    so = d / "libsf3d_hip.so"
    shutil.copy(lib, so)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=d)      # writes <so>.0.hipv4-...gfx950 next to it
    co = [p for p in d.iterdir() if "gfx950" in p.name]
    assert len(co) == 1, [p.name for p in d.iterdir()]
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co[0])], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        out[dem] = dict(lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"), vgpr=g("vgpr_count"),
                        threads=g("max_flat_workgroup_size"))
    assert len(out) > 50
    return out


def _pick(kernels, pattern):
    sel = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert sel, pattern
    return sel


def test_paired_sweeps_keep_the_blocks_per_cu_the_cost_model_counts_on(kernels):
    for W in (6, 10, 14):
        per_cu = 4 * 6 // (W + 1)                 # sf3d_host_build.inc: perCU = 4 * SF3D_PAIR_WAVES / (W + 1)
        # k_sweep_pair<W, NT, DIST, RECORDS> (RECORDS: the strip variant that waits for the neighbours' records inside the launch), k_sweep_pair_masked<W, NT, DIST>
        for name, r in _pick(kernels, rf"k_sweep_pair(_masked<{W}, (true|false), (true|false)>|<{W}, (true|false), (true|false), (true|false)>)").items():
            assert r["threads"] == (W + 1) * 64, (name, r)
            assert r["lds"] * per_cu <= LDS_PER_CU, (name, r, per_cu)
            assert r["vgpr"] <= (VGPRS_PER_SIMD_LANE // 6) // 8 * 8 or per_cu * (W + 1) <= 4 * (VGPRS_PER_SIMD_LANE // r["vgpr"]), (name, r)
            flags = re.search(r"<\d+, (true|false), (true|false)(?:, (true|false))?>", name).groups()
            if flags[1] == "false" or (flags[2] == "false" and "masked" not in name):
                # one GPU - the headline path - and the strip variant with the plain exchange hold no scratch at all (the strip variants of
                # the masked pass keep 24 B, the record hand-over's halo waves 36 B)
                assert r["scratch"] == 0, (name, r)


def test_node_kernels_hold_their_occupancy_without_scratch(kernels):
    budget = {r"k_props<0, false>": 96, r"k_props<2, false>": 96, r"k_post<(true|false)>": 64, r"k_assemble<(true|false), (true|false), false>": 128,
              r"k_restore<(true|false), false>": 112, r"k_sweep<[01], (true|false)>": 64, r"k_sweep<2, (true|false)>": 72}
    for pattern, vg in budget.items():
        for name, r in _pick(kernels, "^void " + pattern + r"\(DevView\)$").items():
            assert r["vgpr"] <= vg, (name, r)
            assert r["scratch"] == 0, (name, r)
