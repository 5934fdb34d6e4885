"""How two oracle instances running side by side (tests/test_gpu_ravone_project.py) share the host: per-step time of one instance on
16 threads alone against two at once (threads of one process, each on a library copy of its own).  Prints the host's CPU limits."""
import os, sys, time, subprocess
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from concurrent.futures import ThreadPoolExecutor
from criteria3d_amd import catchment as cm
from tests import checkers

print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, e)
print(subprocess.run("lscpu | grep -E 'Socket|NUMA|Thread|Core|Model name'", shell=True, capture_output=True, text=True).stdout)
print({k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP", "KMP"))})
m = cm.catchment_model(512, 512, 20)
a, b = checkers.load_oracle(), checkers.load_oracle_copy("b")
def run(sf, threads, steps=6):
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset"); cm.build(sf, m, threads=threads)
    cm.run_hour(sf, m, 20.0, max_steps=2)
    t = time.time(); cm.run_hour(sf, m, 20.0, max_steps=steps); return (time.time() - t) / steps
for th in (8, 16, 32):
    print(f"one instance, {th} threads: {run(a, th):.3f} s/step", flush=True)
for th in (8, 16):
    with ThreadPoolExecutor(2) as pool:
        r = [j.result() for j in [pool.submit(run, sf, th) for sf in (a, b)]]
    print(f"two instances, {th} threads each: {r[0]:.3f} / {r[1]:.3f} s/step", flush=True)
