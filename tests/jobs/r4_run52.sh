set -x
mkdir -p gpurun_out
SOAK_STREAMS=13x100 timeout -k 10 500 python3 tests/micro/soak_pipeline.py 500 11 4 L4 > gpurun_out/r4_soak_13x100.txt 2>&1; echo rc $?; tail -3 gpurun_out/r4_soak_13x100.txt
SOAK_OPTS="large_step_pieces=2" SOAK_STREAMS=13x100 timeout -k 10 500 python3 tests/micro/soak_pipeline.py 500 12 4 L4 > gpurun_out/r4_soak_13x100_p2.txt 2>&1; echo rc $?; tail -3 gpurun_out/r4_soak_13x100_p2.txt
SOAK_STREAMS=13x320 timeout -k 10 500 python3 tests/micro/soak_pipeline.py 150 13 4 L4 > gpurun_out/r4_soak_13x320.txt 2>&1; echo rc $?; tail -3 gpurun_out/r4_soak_13x320.txt
timeout -k 10 500 python3 tests/micro/soak_pipeline.py 3000 14 > gpurun_out/r4_soak_default.txt 2>&1; echo rc $?; tail -3 gpurun_out/r4_soak_default.txt
