cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
COMMON="--no-grouped --no-cpu-baseline --no-extra-configs --no-buffered --no-profile-pass --no-host-pcm --no-f32-engine --no-b512 --regions 3"
NASR_LIB_PATH=$GRAFT_REPO_ROOT/nemotron-asr.cpp_amd/libnemotron_asr_amd_stamps.so NASR_STAMPS_OUT=gpurun_out/r5_stamps_b1.txt python bench.py $COMMON --steps 100 > gpurun_out/r5_stamps_b1.log 2>&1; echo "stamps rc=$?"
MS=$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_stamps_b1.log | head -1 | grep -o '[0-9.]*$')
echo "stamps build ms_per_step=$MS"
python tests/micro/stamps_timeline.py gpurun_out/r5_stamps_b1.txt gpurun_out/r5_b1_R0_pipelined_trace.json $MS > gpurun_out/r5_b1_R0_pipelined_stamps.txt; tail -8 gpurun_out/r5_b1_R0_pipelined_stamps.txt
PROF_COUNTERS="" bash tests/prof_r5.sh b1_R0_pipelined
