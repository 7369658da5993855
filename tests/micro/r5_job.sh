cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu_tests_3.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_gpu_tests_3.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_2.log 2>&1; echo "bench rc=$?"; tail -c 2500 gpurun_out/r5_bench_2.log
