set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0 opt:tile_bands=0" "opt:wide_tiles=3" > gpurun_out/r4_variant_identity_final2.txt 2>&1
echo identity rc $?; cat gpurun_out/r4_variant_identity_final2.txt
NASR_REPORT_DIR=gpurun_out/r4_reports timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite6.txt 2>&1
echo suite rc $?; tail -3 gpurun_out/r4_gpu_suite6.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_final4.json 2> gpurun_out/r4_bench_final4.err
echo bench rc $?
