#!/bin/bash
# round 6 job 46: the fall-back forms on the final code - what bench.py switches to when a first contact fails: the multi-rank test file with the paired pass's
# (without -x: the two cases of test_paired_sweep_on_strips that compare the default against SF3D_PAIR_RECORDS=0 count no difference in mailbox rounds when the
# environment turns the hand-over off for both - their assertion on that count fails by construction; everything else must pass)
# record hand-over off (two launches, plain exchange), and with the resident loop off as well
mkdir -p gpurun_out
( time SF3D_PAIR_RECORDS=0 timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -m gpu ) 2>&1 | grep -v "Gloo\|socket.cpp" | tail -8 | tee gpurun_out/r06_job46_records_off.txt
( time SF3D_PAIR_RECORDS=0 SF3D_RESIDENT_SWEEP=0 timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -m gpu -k "not resident_sweep_loop" ) 2>&1 | grep -v "Gloo\|socket.cpp" | tail -8 | tee gpurun_out/r06_job46_both_off.txt
