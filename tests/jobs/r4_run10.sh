set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_diar.py -m gpu -x -q -k "f32_engine_64 or bf16_token_agreement or borrower" > gpurun_out/r4_new_tests.txt 2>&1
echo new tests rc $?
tail -5 gpurun_out/r4_new_tests.txt
PROF_COUNTERS="" PROF_STEPS=6 timeout -k 10 500 bash tests/prof_r4.sh b512_persist --batch 512 --right-context 13 --weights q8_0 --checkpoint random > gpurun_out/r4_prof_b512_persist.txt 2>&1
PROF_COUNTERS="" PROF_STEPS=6 timeout -k 10 500 bash tests/prof_r4.sh b512_pertile --batch 512 --right-context 13 --weights q8_0 --checkpoint random --engine-option persistent_gemm=0 > gpurun_out/r4_prof_b512_pertile.txt 2>&1
cat gpurun_out/r4_prof_b512_persist.txt gpurun_out/r4_prof_b512_pertile.txt
MARGIN_ALPHAS=0.4 timeout -k 10 400 python3 tests/micro/margin_sweep.py gpu gpurun_out/margin > gpurun_out/r4_margin_gpu_0p4.log 2>&1
echo margin rc $?
(time timeout -k 10 900 python3 bench.py --steps 20 --warmup 5) > gpurun_out/r4_bench_default.txt 2> gpurun_out/r4_bench_default.err
echo bench rc $?
tail -c 3000 gpurun_out/r4_bench_default.txt
tail -5 gpurun_out/r4_bench_default.err
