set -x
mkdir -p gpurun_out
for b in 256 384 512; do AB_SYNC=0 AB_BATCH=$b timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:wide_tiles=3" > gpurun_out/r4_ab_b${b}_mt7pipe.txt 2>&1; echo "== $b"; cat gpurun_out/r4_ab_b${b}_mt7pipe.txt; done
