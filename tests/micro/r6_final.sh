mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r6/gpu_tests_final.log 2>&1; echo "gpu suite rc=$?"; tail -n 14 gpurun_out/r6/gpu_tests_final.log | grep -v amdgpu.ids
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/r6/smoke.log
python bench.py > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; echo "bench rc=$?"
