#!/usr/bin/env python3
"""Expected strong-scaling curve of the headline workload (C4 512 x 512 x 20, F20, the 6-hour episode) at N = 2, 4, 8 GPUs - written
BEFORE any run on more than one physical GPU, so that the driver's first SCALE record has something to be read against.

Inputs (all measured on ONE MI355X; strips: profiles/r06_d_bench_*.json, one box - C4E with the resident sweep loop, csrc/sf3d_resident.inc;
exchange statistics: profiles/r06_d_bench_2ranks_shared.json):
  * T_strip(N): the episode time of one strip of C4 run as a grid of its own (bench.py --workload C4H / C4Q / C4E = one of two / four /
    eight strips; C4 itself for N = 1) - every kernel of the step at the size a rank sees, including whether the paired sweep pays there;
  * E: exchange epochs per episode (sf3d_dist_stats of a two-rank run: one per Jacobi iteration or pair half, per K / waterFlow halo,
    per balance / Courant decision) - independent of N;
  * the extra launches a strip has and a grid of its own has not: k_sweep_bnd per paired pass, two k_halo_copy per approximation.
Assumption (stated, not measured): what ONE exchange epoch costs a rank between physical GPUs - the flag hop over xGMI plus the skew of
the slowest neighbour.  Between two processes on one die the hop is 0.94 us (k_dist_hop) and a rank waits 5.7-6.9 us per epoch on
average, time-slicing included; the table is given for 2, 5 and 10 us.

  T(N) = T_strip(N) + E * epoch_cost + extra_launches * launch_cost          predicted sim-h/s = 6 h / T(N)

What the model cannot know: contention of eight ranks' puts on the xGMI links (160 KB per sweep per rank: negligible against 7 x 153
GB/s), host-side jitter of eight processes polling, and whether device-initiated system-scope stores cross GPUs at all (if not, the
host-memory windows take over at PCIe latency - epoch cost then 10-20 us).

Round 6: at N = 8 the Jacobi iterations of an approximation run inside ONE persistent launch per rank (the resident loop); an exchange epoch there
is a record hand-off between running kernels, not a kernel boundary plus a last-block mailbox round - the same assumed costs are applied to it
(pessimistic), and the k_sweep_bnd launches of the paired pass do not exist at N = 8.  At N = 2 / 4 the paired pass hands its edge rows over as
records while it runs (sf3d_pair.inc, DIST "record hand-over"): no k_sweep_bnd launch and ONE exchange epoch per pass - E minus the passes of an
episode (309: 5 511 against 3 966 mailbox rounds over five episodes, profiles/r06_i_record_handover_on_strips_ab.txt).  At every N the halos of
K and waterFlow are tagged records the reader's copy kernel waits for - no barrier across the ranks behind k_props<2>: one epoch per
approximation less than the two-rank line of job 10 counted (1 103 -> 970).

usage: python scripts/scale_model.py [profiles-dir]  -> profiles/r06_scale_model.json"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
prof = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles"


def line(name):
    return json.loads((prof / name).read_text().strip().splitlines()[-1])


base = line("r06_d_bench_C4.json")
strips = {1: base, 2: line("r06_d_bench_C4H.json"), 4: line("r06_d_bench_C4Q.json"), 8: line("r06_z_bench_C4E.json")}      # (C4E: the final kernels, post-solve part fused)
PASSES = 309                                        # paired passes per episode = exchange epochs the record hand-over saves (see above)
two = line("r06_d_bench_2ranks_shared.json")
E = two["exchange"]["epochs"]                       # per 6-hour episode
work = base["roofline"]["step"]["work"]
E -= work["approximations"]                         # K / waterFlow halos travel as tagged records: no barrier behind k_props<2> (5 511 -> 4 846 rounds over five episodes, job 35)
pairs = base["roofline"]["kernels"]["k_sweep_pair"]["launches"] if "k_sweep_pair" in base["roofline"]["kernels"] else 0
LAUNCH_US = 5.0                                     # a small extra launch inside a replayed graph (profiles/r03_barrier_probe.txt: 1.7-2.7 us boundary + a few us of work)
out = {"workload": "C4 512x512x20, F20, 6-hour episode (bench.py default)", "exchange_epochs_per_episode": E,
       "same_die_measurements": {"flag_hop_us": two["exchange"]["hop_us"], "mean_wait_per_epoch_us": two["exchange"]["mean_wait_us"],
                                 "two_ranks_sharing_one_gpu_sim_h_per_s": two["value"]},
       "inputs": {str(n): {"strip_as_own_grid_sim_h_per_s": s["value"], "episode_ms": 6e3 / s["value"], "dominant_kernel": s["roofline"]["kernel"],
                           "dominant_kernel_avg_us": s["roofline"]["avg_us"]} for n, s in strips.items()},
       "assumed_epoch_cost_us": [2.0, 5.0, 10.0], "predicted": {}}
t1 = 6e3 / strips[1]["value"]          # (`value` = 6 h / median episode; ms_per_step is the measured K-hour figure since round 6)
for n in (2, 4, 8):
    ts = 6e3 / strips[n]["value"]
    paired = strips[n]["roofline"]["kernel"] == "k_sweep_pair"
    # extra launches of a strip: two halo copies per approximation (rounds 3-5 also: k_sweep_bnd per paired pass); epochs: one per pass with the hand-over
    extra = 2 * work["approximations"]
    En = E - PASSES if paired else E
    row = {"compute_only_sim_h_per_s": 6e3 / ts, "compute_only_speedup": t1 / ts, "exchange_epochs": En}
    for e in out["assumed_epoch_cost_us"]:
        t = ts + (En * e + extra * LAUNCH_US) / 1e3
        row[f"epoch_{e:g}us"] = {"episode_ms": t, "sim_h_per_s": 6e3 / t, "speedup": t1 / t, "efficiency": t1 / t / n}
    out["predicted"][str(n)] = row
(prof / "r06_scale_model.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out["predicted"], indent=1))
