set -x
mkdir -p gpurun_out
timeout -k 10 700 python3 tests/server_load.py --streams 256 --seconds 30 --right-context 13 --mode burst --client native --warmup-seconds 4 --conns 16 --workdir gpurun_out/srv256 > gpurun_out/r4_server_256_burst.json 2> gpurun_out/r4_server_256_burst.err
echo rc $?
tail -c 1800 gpurun_out/r4_server_256_burst.json; tail -5 gpurun_out/r4_server_256_burst.err
