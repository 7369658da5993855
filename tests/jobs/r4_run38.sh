set -x
mkdir -p gpurun_out
PROF_STEPS=6 timeout -k 10 500 bash tests/prof_r4.sh b512_R13 --batch 512 --right-context 13 --weights q8_0 --checkpoint random > gpurun_out/r4_prof_b512_R13.txt 2>&1
tail -25 gpurun_out/r4_prof_b512_R13.txt
PROF_STEPS=6 timeout -k 10 500 bash tests/prof_r4.sh b512_R13_rowsfirst --batch 512 --right-context 13 --weights q8_0 --checkpoint random --engine-option tile_bands=0 > gpurun_out/r4_prof_b512_R13_rowsfirst.txt 2>&1
tail -25 gpurun_out/r4_prof_b512_R13_rowsfirst.txt
