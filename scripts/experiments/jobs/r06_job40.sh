#!/bin/bash
# round 6 job 40: the opt-in long tests on the final code (SF3D_FULL_MATRIX=1: every switchable launch form on its own x three cases, bit for bit;
# SF3D_LONG_TESTS=1: the long runoff run, the long project-window run with its Ravone window, twelve coupled water + heat hours against the oracle)
mkdir -p gpurun_out
( time SF3D_FULL_MATRIX=1 SF3D_LONG_TESTS=1 timeout 1500 python -m pytest tests/ -x -q -m gpu -k "launch_mode or long or twelve or full_matrix or restore_best or runoff_regime or coupled_heat_hours or window1 or heat_hours" --durations=8 ) > gpurun_out/r06_job40_long.log 2>&1; tail -16 gpurun_out/r06_job40_long.log
