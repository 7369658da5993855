set -x
mkdir -p gpurun_out
timeout -k 10 500 bash tests/prof_r4.sh b1_R0 > gpurun_out/r4_prof_b1_R0.txt 2>&1
timeout -k 10 600 bash tests/prof_r4.sh b64_R13 --batch 64 --right-context 13 --weights q8_0 > gpurun_out/r4_prof_b64_R13.txt 2>&1
PROF_COUNTERS="" PROF_STEPS=6 timeout -k 10 400 bash tests/prof_r4.sh b512_R13 --batch 512 --right-context 13 --weights q8_0 --checkpoint random > gpurun_out/r4_prof_b512_R13.txt 2>&1
PROF_COUNTERS="" PROF_STEPS=10 timeout -k 10 400 bash tests/prof_r4.sh f32_b64_R13 --batch 64 --right-context 13 --dtype f32 --checkpoint random > gpurun_out/r4_prof_f32_b64_R13.txt 2>&1
cat gpurun_out/r4_prof_b1_R0.txt gpurun_out/r4_prof_b64_R13.txt gpurun_out/r4_prof_b512_R13.txt gpurun_out/r4_prof_f32_b64_R13.txt
ls gpurun_out/*kernel_stats* gpurun_out/r4_pmc*
