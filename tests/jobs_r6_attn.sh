set -x
mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q -k "512 or 256" > gpurun_out/r6/t_attn.log 2>&1; echo "tests rc=$?"
timeout -k 10 600 python tests/micro/gemm_variant_identity.py > gpurun_out/r6/identity2.log 2>&1; echo "identity rc=$?"
python bench.py > gpurun_out/r6/bench_b.json 2> gpurun_out/r6/bench_b.err; echo "bench rc=$?"
