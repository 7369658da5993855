set -x
mkdir -p gpurun_out
timeout -k 10 1000 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0 opt:tile_bands=0" "opt:wide_tiles=256 opt:tile_bands=1" "opt:wide_tiles=3" "opt:large_step_pieces=0" > gpurun_out/r4_variant_identity_final.txt 2>&1
echo identity rc $?
cat gpurun_out/r4_variant_identity_final.txt
