mkdir -p gpurun_out/final_k
python -m pytest tests -m gpu -x -q > gpurun_out/final_k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_k/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_k/smoke.log 2>&1
python bench.py > gpurun_out/final_k/bench.json 2> gpurun_out/final_k/bench.err
python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/final_k/bench_untimed.json 2> /dev/null
bash scripts/profile_gpu.sh r01k --time-all-kernels > gpurun_out/final_k/profile.log 2>&1
tail -3 gpurun_out/final_k/pytest.log; cat gpurun_out/final_k/smoke.log | tail -1; cat gpurun_out/final_k/bench.json | cut -c1-200
