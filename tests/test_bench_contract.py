"""The one JSON line bench.py prints is a contract with the driver: this checks the committed line of the last profiled
run (profiles/*_bench.json, produced by `python bench.py` on an MI355X) for every field the contract names."""
import glob
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_committed_bench_line_has_every_contract_field():
    files = sorted(glob.glob(str(ROOT / "profiles" / "r0?_?_bench.json")))
    assert files, "no committed bench line under profiles/"
    line = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"].startswith("simulated-hours/sec on 512x512x20") and line["unit"] == "sim-h/s"
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    if Path(files[-1]).name >= "r06":
        # round 6 on: `ms_per_step` is the MEASURED figure (the K timed hours over their time, median repetition) - `value` stays the 6-hour episode;
        # the line carries its own parity evidence and the other BASELINE configs (config 5 water / + heat, config 3 F60) timed on the same box
        assert abs(line["ms_per_step"] - line["timed_region"]["elapsed_s"] / line["steps"] * 1e3) < 1e-9 * line["ms_per_step"]
        assert "bit-identical" in line["parity"]["c2_f20_hour0_vs_reference_vector"]
        legs = line["legs"]
        for name, nodes in (("c5_hour0", 5845035), ("c5_heat_hour0", 5845035), ("c3_f60_2h", 983040)):
            leg = legs[name]
            assert leg["unit"] == "sim-h/s" and leg["value"] > 0 and leg["nodes"] == nodes and leg["work"]["accepted"] > 0, name
            assert leg["dominant_kernel"]["avg_us"] > 0 and 0 < leg["dominant_kernel"]["frac"] <= 1 and 0 < leg["step"]["frac"] <= 1, name
        assert legs["c5_heat_hour0"]["step"]["heat_work"]["accepted"] > legs["c5_heat_hour0"]["work"]["accepted"]      # several heat sub-steps per water step
        assert line["roofline"]["traffic_is_of_this_build"] is True      # the stored PMC profile was taken on the kernel sources of this line's build
    else:
        assert abs(line["value"] - line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if Path(files[-1]).name >= "r03":
        # round 3 on: `frac` is a PHYSICAL fraction - the bytes one launch has to move over its duration - never above 1; the
        # two-sweeps-per-launch pricing of the paired sweep lives in `equivalent_sweep_frac`; the whole timed region and the C4 F60
        # hour 0 (runoff regime) are on the same line
        assert 0.3 < r["frac"] <= 1.0, r["frac"]
        assert r["kernel"] != "k_sweep_pair" or (r["equivalent_sweep_frac"] > r["frac"] and r["algorithmic_bytes_per_launch"] == 160 * 5242880)
        st = r["step"]
        assert 0.2 < st["frac"] <= 1.0 and st["bytes"] > 0 and st["elapsed_s"] > 0 and st["survey_8d_frac"] >= st["frac"]
        f = line["f60_hour0"]
        assert f["unit"] == "sim-h/s" and f["value"] > 0 and f["work"]["accepted"] == 76 and f["work"]["courant_rejections"] == 42
        assert "value_timing" in line
        if Path(files[-1]).name >= "r04":
            # round 4 on: `value` is the 6-hour episode whatever --steps is, the step roofline describes one episode
            assert line["headline_6h"]["value"] == line["value"] and line["timed_region"]["hours"] == line["steps"]
            assert st["region"].startswith("one 6-hour episode") and st["work"]["sweeps"] == 703 and st["traffic"] is not None
            # the driver's own command (--steps 20 --warmup 5), latest committed line of the same round
            drv_files = sorted(glob.glob(str(ROOT / "profiles" / (Path(files[-1]).name[:3] + "_*_bench_driver_style_steps20_warmup5.json"))))
            assert drv_files, "no committed driver-style bench line for this round"
            drv = json.load(open(drv_files[-1]))
            # two runs on one box differ by up to 4 % (round 6: the pass at 152.8 against 141.5 us minutes apart), hence 5 %
            assert drv["steps"] == 20 and abs(drv["value"] - line["value"]) < 0.05 * line["value"], (drv["value"], line["value"])
    else:
        assert 0.5 < r.get("pass_frac", r["frac"]) < 1.0
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "sim-h/s" and c["cores"] >= 1 and c["value"] > 0


import pytest


@pytest.mark.gpu
def test_bench_runs_and_prints_one_json_line():
    """a short real run (C2, one simulated hour, CPU baseline on a 3 s budget): exactly one JSON line on stdout"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "C2", "--steps", "1", "--warmup", "1", "--cpu-budget", "3"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["warmup"] == 1 and line["value"] > 0
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["kernel"] == "k_sweep_resident"      # (C2's rows fit on chip: all iterations of an approximation in one launch)
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["value"] > 0
    assert line["config"]["work"]["accepted"] == 22          # C2 F20 hour 0


def _run_shared_gpu_bench(cmd, env):
    """bench.py with its ranks on the ONE GPU of the test box.  Several processes' spinning kernels on one device depend on the hardware
    scheduler running them side by side (tests/test_gpu_multirank.py::run_ranks): a SET-UP time-out (exit code 4, "strips not
    connected") is retried once, loudly; any other failure is final."""
    import subprocess
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if p.returncode == 4 and "strips not connected" in p.stderr and "SF3D_BENCH_FORCE_REPLICAS" not in env:
        import warnings
        warnings.warn("bench.py: the ranks sharing the GPU could not connect their strips, retrying once:\n" + p.stderr[-1500:])
        if "--master-port" in cmd:
            cmd = list(cmd); cmd[cmd.index("--master-port") + 1] = str(int(cmd[cmd.index("--master-port") + 1]) + 400)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    return p


@pytest.mark.gpu
@pytest.mark.gpu_timing          # (rank processes of its own on the shared GPU: after the background worker's product runs, like the timing tests)
def test_bench_two_ranks_sharing_the_gpu():
    """the launch the driver uses for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with both ranks on the
    one GPU of the test box (SF3D_BENCH_SHARE_GPU=1: gloo control plane, same device-side exchange)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SF3D_BENCH_SHARE_GPU="1")
    p = _run_shared_gpu_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", "29631", str(ROOT / "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "1", "--warmup", "0",
                               "--no-cpu-baseline"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["work"]["accepted"] == 22
    # the roofline of an N > 1 line prices what ONE rank's launch works on: the nodes it owns (C2 in two strips: 20 480 of 40 960), once
    r = line["roofline"]
    # (the resident sweep loop: 104 B/node once + 16 B/node per iteration, 4-6 iterations per launch)
    assert r["kernel"] == "k_sweep_resident" and 150 * 20480 < r["algorithmic_bytes_per_launch"] < 220 * 20480, r
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9) < 1e-9 * r["achieved"]
    assert "exchange HIP-IPC windows - windows passed the self-check" in p.stderr and "SAME GPU" in p.stderr      # every rank says which exchange came up
    # ... and what it cost: transport, flag-hop latency to the peer, wait per exchange epoch - per rank on stderr, rank 0's on the line
    import re
    for r in (0, 1):
        mt = re.search(rf"\[bench\] rank {r}: exchange transport HIP-IPC device windows; peer {1 - r}: hop ([0-9.]+) us, wait per epoch mean ([0-9.]+) us max ([0-9.]+) us; (\d+) exchange epochs", p.stderr)
        assert mt, p.stderr[-3000:]
        hop, mean, mx, epochs = float(mt.group(1)), float(mt.group(2)), float(mt.group(3)), int(mt.group(4))
        assert 0.2 < hop < 500 and mean > 0 and mx >= mean and epochs > line["config"]["work"]["sweeps"]
    assert "flag hop through the windows" in p.stderr
    ex = line["exchange"]
    assert ex["transport"] == "HIP-IPC device windows" and set(ex["hop_us"]) == {"1"} and ex["epochs"] > 0


@pytest.mark.gpu
@pytest.mark.gpu_timing
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher - how the driver calls it - starts two fresh rank processes itself, relays rank
    0's line and carries the self-describing fields (driver-timed 6-hour headline, repetitions, inclusive rate, traffic source)"""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SF3D_BENCH_SHARE_GPU"] = "1"
    p = _run_shared_gpu_bench([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "6", "--warmup", "0", "--reps", "2",
                               "--no-cpu-baseline"], env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0
    assert len(line["repeats_s"]) == 2 and min(line["repeats_s"]) > 0          # --reps 2: rewound, not rebuilt, and checked to do the same work
    assert line["headline_6h"]["value"] == line["value"]      # `value` IS the 6-hour episode ...
    assert abs(line["ms_per_step"] - line["timed_region"]["elapsed_s"] / 6 * 1e3) < 1e-9 * line["ms_per_step"]      # ... `ms_per_step` the measured hours (round 6)
    # an N > 1 line validates itself: the reference's vector in two strips, and the parity keys travel on the line
    assert "2 strips" in line["parity"]["c2_f20_hour0_vs_reference_vector"] and line["exchange"]["parity"] == line["parity"]
    assert "parity ok: C2 F20 hour 0 in 2 strip(s)" in p.stderr
    assert line["timed_region"]["hours"] == 6 and line["roofline"]["step"]["region"].startswith("one 6-hour episode")
    assert 0 < line["inclusive_value"] <= line["value"] * 1.02
    assert line["config"]["work"]["accepted"] == 50          # C2 F20: 22 + 13 + 6 + 3 + 3 + 3
    assert "traffic_source" in line["roofline"]


@pytest.mark.gpu
@pytest.mark.gpu_timing          # (compares two timings: run when nothing else is on the GPU - tests/conftest.py moves it behind the background worker's GPU phase)
def test_bench_value_is_the_six_hour_episode_whatever_steps_is():
    """the driver runs `--steps 20 --warmup 5`: the timed hours walk through the 6-hour episode of SURVEY 8d again and again (rewound
    outside the clock), `value` is 6 h over the median COMPLETE episode - the 8d headline - and does not depend on --steps"""
    import subprocess
    import sys
    out = {}
    for steps, warm in ((20, 5), (6, 1)):
        p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "C2", "--steps", str(steps), "--warmup", str(warm), "--reps", "3",
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        out[steps] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    a, b = out[20], out[6]
    assert a["steps"] == 20 and a["warmup"] == 5 and a["headline_6h"]["value"] == a["value"]
    # two separate runs of a launch-bound grid (0.26 ms per computeStep, the host's launch rate decides): 424 / 435 / 454 sim-h/s were
    # measured for --steps 20 / 6 / 12 on one box, 444 against 394 on another while the suite's background worker kept the host cores
    # busy - run-to-run noise; what this guards against is the round-3 defect (K hours over their time: 2.6 x between --steps 20 and 6);
    # the committed C4 lines (--steps 6 and --steps 20 --warmup 5) are held to 3 % in the CPU test above
    assert abs(a["value"] - b["value"]) < 0.25 * b["value"], (a["value"], b["value"])
    assert len(a["headline_6h"]["episodes_s"]) == 3 * 3 and a["timed_region"]["hours"] == 20
    assert a["config"]["work"]["accepted"] == 3 * 50 + 22 + 13          # three episodes (22 + 13 + 6 + 3 + 3 + 3) and hours 0, 1 of a fourth
    assert a["roofline"]["step"]["work"]["accepted"] == 50               # the step roofline describes ONE episode


@pytest.mark.gpu
def test_bench_does_not_turn_a_failed_exchange_into_a_headline():
    """strips that cannot be connected (forced here) are an ERROR: exit code 4, the reason of every rank on stderr, no JSON line.
    Only SF3D_BENCH_ALLOW_REPLICAS=1 runs N labelled replicas - and then `value` is ONE replica's rate, never N times it"""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SF3D_BENCH_ALLOW_REPLICAS")}
    env["SF3D_BENCH_SHARE_GPU"] = "1"; env["SF3D_BENCH_FORCE_REPLICAS"] = "1"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "0", "--reps", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")] and "strips not connected" in p.stderr
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, SF3D_BENCH_ALLOW_REPLICAS="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "INDEPENDENT REPLICAS" in line["config"]["partition"]
    assert abs(line["value"] - 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]          # two hours over one replica's time (K < 6: `value` is the K hours over their time)
    assert abs(line["replica_throughput"] - 2 * line["value"]) < 1e-9 * line["value"]
    assert line["config"]["work"]["accepted"] == 22 + 13


@pytest.mark.gpu
def test_bench_rank_without_a_gpu_stops_the_run():
    """two ranks asked for on a box with one GPU (and no SF3D_BENCH_SHARE_GPU): rank 1 has no device of its own and says so; the
    parent stops rank 0 instead of leaving it waiting in a collective, and returns the failing rank's code"""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with ONE GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SF3D_BENCH_SHARE_GPU")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    assert "no GPU of its own" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_source_compiles_without_warnings():
    """a SyntaxWarning in bench.py ("'str' object is not callable; perhaps you missed a comma?") is a run-time TypeError on the GPU box:
    compile the file with warnings as errors here, where no GPU is needed"""
    import warnings
    src = (ROOT / "bench.py").read_text()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        compile(src, "bench.py", "exec")


def test_bench_refuses_a_world_size_mismatch():
    """--gpus must equal the launcher's WORLD_SIZE; the message says what to do (runs without a GPU: the check comes first)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_bench_reads_an_existing_pmc_summary():
    """roofline.traffic comes from the committed rocprofv3 PMC summary bench.py names: the file must exist and carry the dominant kernel"""
    import re
    src = (ROOT / "bench.py").read_text()
    tags = re.search(r'for tag in \(([^)]*)\)', src).group(1)
    names = [f"{t.strip().strip(chr(34))}_kernel_summary.json" for t in tags.split(",")]
    name = next(n for n in names if (ROOT / "profiles" / n).exists())
    prof = json.load(open(ROOT / "profiles" / name))
    sweep = "k_sweep_pair" if "k_sweep_pair" in prof else "k_sweep"      # the paired sweep is the dominant kernel since round 2
    assert prof[sweep]["hbm_traffic_MB"] > 0.5 * (160 if sweep == "k_sweep_pair" else 152) * 5242880 / 1e6
