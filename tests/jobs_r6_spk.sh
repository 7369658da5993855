set -x
mkdir -p gpurun_out/r6
timeout -k 10 300 python tests/micro/spk_parity.py > gpurun_out/r6/spk_parity.log 2>&1; echo "spk_parity rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_diar.py -x -q > gpurun_out/r6/t_diar.log 2>&1; echo "diar tests rc=$?"
bash tests/prof_diar.sh > gpurun_out/r6/diar_kernels_new.txt 2>&1; echo "diar prof rc=$?"
