import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import pathlib, tempfile
import test_gpu_multirank as T
PREV = "/root/repo/build_variants/libsf3d_prev.so"
for envx in ({"SF3D_PAIR_W": "6", "SF3D_NT_STREAM": "1"}, {"SF3D_PAIR_W": "6", "SF3D_NT_STREAM": "1", "SF3D_PRODUCT_LIB": PREV}, {"SF3D_PAIR_W": "6", "SF3D_NT_STREAM": "1", "SF3D_PAIR_RECORDS": "0"},
             {"SF3D_PAIR_W": "6"}, {"SF3D_PAIR_W": "6", "SF3D_NT_STREAM": "1"}):
    for world, case in ((3, "c2f60"),):
        tmp = pathlib.Path(tempfile.mkdtemp())
        try:
            pair = T.run_ranks(world, case, tmp, 29811, env={"SF3D_PAIR_SWEEP": "1", "SF3D_RESIDENT_SWEEP": "0", **envx})
            single = T.run_ranks(world, case, tmp, 29831, env={"SF3D_PAIR_SWEEP": "0", "SF3D_RESIDENT_SWEEP": "0"})
            owner = pair[0]["owner"]; bad = []
            for r in range(world):
                mine = owner == r
                for k in pair[r].files:
                    if k.startswith(("H_h", "Se_h")) and not np.array_equal(pair[r][k][mine], single[r][k][mine]): bad.append((r, k, int((pair[r][k][mine] != single[r][k][mine]).sum())))
            print(envx, world, case, "passes", [int(p["sweep_launches"][1]) for p in pair], "BAD" if bad else "ok", bad[:4], flush=True)
        except AssertionError as e:
            print(envx, world, case, "run failed", str(e)[-300:], flush=True)
