mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r6/gpu_tests_full.log 2>&1; echo "gpu suite rc=$?"
tail -n 25 gpurun_out/r6/gpu_tests_full.log
