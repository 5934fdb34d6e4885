"""One rank of a multi-rank run of the HIP product (one process per rank; all ranks may share one
GPU for functional tests).  The control plane is torch.distributed with the gloo backend.
usage: python scripts/multirank_worker.py <rank> <world> <port> <case> <outfile>"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch.distributed as dist
from criteria3d_amd import capi, catchment as cm

rank, world, port, case, outfile = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("gloo", rank=rank, world_size=world)

def allgather(b):
    out = [None] * world
    dist.all_gather_object(out, b)
    return out

sf = capi.load_product()
if case.endswith("_nofinalize"):
    sf.legacy_connect = True; case = case[:-len("_nofinalize")]
edit_after = case.endswith("_editafter")
if edit_after:
    case = case[:-len("_editafter")]
sf.check(sf.lib.sf3d_set_device(int(os.environ.get("SF3D_TEST_DEVICE", "0"))), "set_device")
if case == "c2f20":
    m, plan = cm.catchment_model(64, 64, 10), [20.0, 0.0]
elif case == "c2f60":
    m, plan = cm.catchment_model(64, 64, 10), [60.0, (0.0, 150)]
elif case == "tall":             # twenty layers on a narrow grid: two ranks of eight rows each
    m, plan = cm.catchment_model(64, 16, 20, heterogeneous=True), [20.0, (0.0, 60)]
elif case == "het64":            # twelve soils on a grid the regular-grid sweeps accept (64 columns), cut into strips of 16 rows
    m, plan = cm.catchment_model(64, 64, 6, heterogeneous=True), [20.0, (0.0, 100)]
elif case == "het":
    m, plan = cm.catchment_model(48, 40, 6, heterogeneous=True), [20.0, (0.0, 150)]
elif case == "ragged":
    m, plan = cm.ragged_model(9, 24, 4), [10.0, 0.0]
elif case == "ravone":           # BASELINE config 5 as specified: the Ravone project (criteria3d_amd/project3d.py), 5.85 M nodes
    from tests.scenarios import ravone_project_model
    m, plan = ravone_project_model(None), [(20.0, 3)]
elif case == "random":
    m, plan = cm.random_model(17, nx=12, ny=40, nz=5), [12.0, (0.0, 30)]
elif case == "c4f20h0":           # BASELINE config 4's cut: 512 x 512 x 20 in 8 row strips, hour 0 of F20 (22 steps)
    m, plan = cm.catchment_model(512, 512, 20), [20.0]
elif case == "holes":             # a layered MASKED grid: random holes, columns of different depth, a random subset of the lateral links
    m, plan = cm.random_model(23, nx=70, ny=45, nz=5), [(30.0, 120), (0.0, 60)]
elif case == "projwin":           # a window of the Ravone project: DEM outline, four soils, short columns
    from tests.scenarios import ravone_project_model
    m, plan = ravone_project_model((980, 1060, 330, 420)), [(25.0, 150), (0.0, 60)]
elif case == "heat":
    m, plan = cm.with_heat_surface(cm.catchment_model(40, 48, 6, heterogeneous=True)), [4.0, 0.0]
else:
    raise SystemExit("unknown case")
heat = cm.Heat(water=True, latent=True, save_mode=0) if case == "heat" else None
sparse = os.environ.get("SF3D_TEST_SPARSE_BUILD") == "1"      # strip-local build: this rank stages its strip and the ring of columns around it only
# a throw-away model first: re-initialisation must drop the windows, re-export and re-connect
sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
if case not in ("ravone", "c4f20h0"):
    cm.build(sf, m, threads=1, dist=(rank, world, allgather), heat=heat, sparse=sparse)
    cm.run_hour(sf, m, 5.0, max_steps=2)
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
t_build = time.time()
cm.build(sf, m, threads=1, dist=(rank, world, allgather), heat=heat, sparse=sparse)
t_build = time.time() - t_build
if edit_after:
    # a graph-dirtying setter AFTER connect / finalize (the strip's build arrays are trimmed by then): every rank must get the
    # documented TopographyError from the next device call - no crash, no peer left in a time-out - and a NaN from computeStep
    cm.run_hour(sf, m, 5.0, max_steps=2)
    codes = {"set": int(sf.lib.sf3d_set_surface_properties(0, 0.05)), "balance": int(sf.lib.sf3d_initialize_balance())}
    dt = float(sf.lib.sf3d_compute_step(10.0))
    np.savez(outfile, set=np.array(codes["set"]), balance=np.array(codes["balance"]), dt=np.array(dt))
    dist.barrier()
    sf.lib.sf3d_clean()
    dist.destroy_process_group()
    sys.exit(0)
owner = sf.owner_map(world, m.n)
res = {"owner": owner, "transport": np.array(int(sf.lib.sf3d_dist_transport())), "host_bytes": np.array(int(sf.lib.sf3d_host_bytes()), dtype=np.int64)}      # the library's resident staging memory once connected
t0 = time.time()
for h, item in enumerate(plan):
    mm, mx = item if isinstance(item, tuple) else (item, None)
    if heat is not None:
        cm.apply_heat_forcing(sf, m, h)
    steps, dts = cm.run_hour(sf, m, mm, max_steps=mx)
    s = cm.snapshot(sf, m)
    if heat is not None:
        res[f"T_h{h}"] = sf.temperature(0, m.n)
    res[f"dts_h{h}"] = np.array(dts)
    res[f"H_h{h}"] = s["H"]; res[f"Se_h{h}"] = s["Se"]
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        res[f"{k}_h{h}"] = np.array(s[k])
res["seconds"] = np.array(time.time() - t0)
res["build_seconds"] = np.array(t_build)
# peak resident set of THIS process image (model arrays of the caller included).  VmHWM belongs to the address space - which exec
# replaces - while getrusage's ru_maxrss is inherited from the parent at fork (a 20 GB pytest session reads as 20 GB in every child)
res["maxrss_mb"] = np.array(next(int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmHWM:")) // 1024)
res["device_bytes"] = np.array(int(sf.lib.sf3d_device_bytes()), dtype=np.int64)
res["host_bytes_end"] = np.array(int(sf.lib.sf3d_host_bytes()), dtype=np.int64)
c = sf.counters()
res["counters"] = np.array([c[k] for k in capi.COUNTER_NAMES[:7]], dtype=np.int64)
res["sweep_launches"] = np.array(sf.sweep_launches() + (sf.resident_launches(),), dtype=np.int64)          # (single sweeps, paired passes, resident loops) of this rank
try:
    res["epochs"] = np.array(sf.dist_stats(world)["epochs"], dtype=np.int64)      # exchange epochs (mailbox rounds) of the whole run
except Exception:
    res["epochs"] = np.array(-1, dtype=np.int64)
np.savez(outfile, **res)
dist.barrier()
sf.lib.sf3d_clean()
dist.destroy_process_group()
