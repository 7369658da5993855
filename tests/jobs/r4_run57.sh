set -x
mkdir -p gpurun_out /tmp/srv256
timeout -k 10 400 python3 tests/server_load.py --streams 256 --seconds 12 --right-context 0 --mode realtime --client native --warmup-seconds 3 --conns 16 --workdir /tmp/srv256 > gpurun_out/r4_server_256_realtime_R0.json 2> gpurun_out/r4_server_256_realtime_R0.err
echo rc $?
tail -c 1500 gpurun_out/r4_server_256_realtime_R0.json; tail -3 gpurun_out/r4_server_256_realtime_R0.err
timeout -k 10 400 python3 tests/server_load.py --streams 256 --seconds 16 --right-context 13 --mode realtime --client native --warmup-seconds 4 --conns 16 --workdir /tmp/srv256 > gpurun_out/r4_server_256_realtime_R13.json 2> gpurun_out/r4_server_256_realtime_R13.err
echo rc $?
tail -c 1500 gpurun_out/r4_server_256_realtime_R13.json; tail -3 gpurun_out/r4_server_256_realtime_R13.err
