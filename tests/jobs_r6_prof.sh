# rocprofv3 evidence for the round-6 build: the headline configuration and configs[2], synchronous steps (kernels alone on the chip: the regime the roofline block prices)
PROF_MODE=--sync-steps bash tests/prof_r6.sh b1_R0 > gpurun_out/r6_prof_b1.txt 2>&1; echo "b1 rc=$?"; tail -n 4 gpurun_out/r6_prof_b1.txt
PROF_MODE=--sync-steps PROF_STEPS=12 bash tests/prof_r6.sh b64_R13 --batch 64 --right-context 13 --weights q8_0 > gpurun_out/r6_prof_b64.txt 2>&1; echo "b64 rc=$?"; tail -n 4 gpurun_out/r6_prof_b64.txt
ls gpurun_out/r6_*
