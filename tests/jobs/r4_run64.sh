set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0 opt:tile_bands=0 opt:t64_tiles=127" "opt:t64_tiles=0" > gpurun_out/r4_variant_identity_t64.txt 2>&1
echo identity rc $?; cat gpurun_out/r4_variant_identity_t64.txt
NASR_REPORT_DIR=gpurun_out/r4_reports timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite7.txt 2>&1
echo suite rc $?; tail -3 gpurun_out/r4_gpu_suite7.txt
