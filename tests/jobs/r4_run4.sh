set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f32" > gpurun_out/r4_f32_tests.txt 2>&1
echo f32 tests rc $?
common="--batch 64 --right-context 13 --dtype f32 --steps 10 --warmup 3 --regions 3 --no-b512 --no-f32-engine --no-host-pcm --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --checkpoint random"
timeout -k 10 600 python3 bench.py $common > gpurun_out/r4_f32_b64_pipelined.txt 2>&1
timeout -k 10 600 python3 bench.py $common --sync-steps > gpurun_out/r4_f32_b64_sync.txt 2>&1
tail -c 600 gpurun_out/r4_f32_b64_sync.txt
