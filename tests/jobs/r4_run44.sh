set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "piece_counts" > gpurun_out/r4_pipe_tests2.txt 2>&1; tail -4 gpurun_out/r4_pipe_tests2.txt
NASR_PROBE_NO_NSEG_DRAIN=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "piece_counts" > gpurun_out/r4_pipe_tests3.txt 2>&1; tail -4 gpurun_out/r4_pipe_tests3.txt
