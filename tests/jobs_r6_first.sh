set -x
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_round6.py -x -q -s > gpurun_out/r6/t_round6.log 2>&1; echo "round6 rc=$?" 
python -m pytest tests/test_gpu_round5.py -x -q -k "stale or cap" > gpurun_out/r6/t_round5_subset.log 2>&1; echo "round5 subset rc=$?"
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r6/t_parity.log 2>&1; echo "parity rc=$?"
python tests/micro/gemm_variant_identity.py > gpurun_out/r6/identity.log 2>&1; echo "identity rc=$?"
bash tests/prof_diar.sh > gpurun_out/r6/diar_kernels.txt 2>&1; echo "diar prof rc=$?"
tail -5 gpurun_out/r6/t_round6.log gpurun_out/r6/t_round5_subset.log gpurun_out/r6/t_parity.log gpurun_out/r6/identity.log
