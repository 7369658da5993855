set -x
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_configs.py -x -q --durations=25 -k "round6 or tolerance or config3 or config4 or 512_streams_R13_one or nan or hazard" > gpurun_out/r6/t_tol.log 2>&1; echo "rc=$?"
