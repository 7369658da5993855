set -x
mkdir -p gpurun_out
PROF_STEPS=16 timeout -k 10 700 bash tests/prof_r4.sh b64_R13 --batch 64 --right-context 13 --weights q8_0 > gpurun_out/r4_prof_b64_R13.txt 2>&1
cat gpurun_out/r4_prof_b64_R13.txt
bash tests/micro/prof_pipelined_b64.sh > gpurun_out/r4_prof_b64_pipelined.txt 2>&1
tail -20 gpurun_out/r4_prof_b64_pipelined.txt
