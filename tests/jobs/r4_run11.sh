set -x
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export NASR_LIB_PATH=$PWD/nemotron-asr.cpp_amd/libnasr_base_ab.so; else unset NASR_LIB_PATH; fi
    echo "== $lib $rep" >> gpurun_out/r4_ab_sigmoid.txt
    timeout -k 10 400 bash tests/micro/ab_b64.sh >> gpurun_out/r4_ab_sigmoid.txt 2>&1
  done
done
cat gpurun_out/r4_ab_sigmoid.txt
unset NASR_LIB_PATH
timeout -k 10 600 python3 -m pytest tests/test_gpu_speech.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4_sigmoid_tests.txt 2>&1
echo tests rc $?
tail -3 gpurun_out/r4_sigmoid_tests.txt
