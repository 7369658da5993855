set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_host_gguf.py -m gpu -x -q -k "language_switch or server" > gpurun_out/r4_lang_test.txt 2>&1
echo rc $?
tail -30 gpurun_out/r4_lang_test.txt | cut -c1-300
