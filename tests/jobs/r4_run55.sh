set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py --gpus 2 --share-device 0 --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline > gpurun_out/r4_two_ranks_final.txt 2> gpurun_out/r4_two_ranks_final.err
echo rc $?
tail -c 1500 gpurun_out/r4_two_ranks_final.txt
NASR_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline --no-extra-configs > gpurun_out/r4_force_dist_final.txt 2>&1
echo rc $?
