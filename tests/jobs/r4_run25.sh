set -x
mkdir -p gpurun_out
NASR_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline > gpurun_out/r4_forcedist.txt 2> gpurun_out/r4_forcedist.err
echo rc $?
tail -c 1500 gpurun_out/r4_forcedist.txt
tail -3 gpurun_out/r4_forcedist.err
