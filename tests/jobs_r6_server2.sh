set -x
mkdir -p gpurun_out/r6/server
timeout -k 10 600 python -m pytest tests/test_gpu_server_load.py tests/test_host_gguf.py -x -q -m gpu > gpurun_out/r6/t_server.log 2>&1; echo "server tests rc=$?"
for R in 13 0; do
timeout -k 10 300 python tests/server_load.py --streams 64 --seconds 12 --right-context $R --mode realtime --client native --conns 8 --warmup-seconds 3 > gpurun_out/r6/server/live_R${R}_64.json 2> gpurun_out/r6/server/live_R${R}_64.err; echo "live R=$R rc=$?"
done
nproc; cat /proc/cpuinfo | grep -c processor; python -c "import os; print(len(os.sched_getaffinity(0)))"
