#!/bin/bash
# round 5 job 33: the counters of round 2's deep dive (profiles/r02_c_pmc_deep_dive_pair_assemble.json) for the gather-free k_sweep_pair: what does
# the pass look like now - L1 -> L2 requests and their latency, instruction mix, wait cycles (one --pmc pass per group, kernel trace only beside it)
mkdir -p gpurun_out
{
bash scripts/pmc_probe.sh r05yA "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
bash scripts/pmc_probe.sh r05yB "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
bash scripts/pmc_probe.sh r05yC "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC"
bash scripts/pmc_probe.sh r05yD "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
} > gpurun_out/r05_job33_pmc_deep_dive.txt 2>&1
grep -c . gpurun_out/r05_job33_pmc_deep_dive.txt; grep "k_sweep_pair" gpurun_out/r05_job33_pmc_deep_dive.txt | head -40
find gpurun_out/r05yA gpurun_out/r05yB gpurun_out/r05yC gpurun_out/r05yD -name "*.csv" -size +4M -delete
