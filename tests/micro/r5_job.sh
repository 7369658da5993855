cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tests/micro/gemm_variant_identity.py > gpurun_out/r5_identity.log 2>&1; echo "identity rc=$?"; tail -2 gpurun_out/r5_identity.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f32_engine_matches or bf16_engine_matches or large_m or quantised" > gpurun_out/r5_par.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r5_par.log
C512="--batch 512 --right-context 13 --weights q8_0 --no-grouped --no-cpu-baseline --no-extra-configs --no-buffered --no-host-pcm --no-f32-engine --no-b512 --steps 20 --warmup 4 --regions 3"
timeout -k 10 300 python bench.py $C512 > gpurun_out/r5_b512.log 2>&1; echo "b512 rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_b512.log | head -1) sync $(grep -o '"synchronous_ms_per_step": [0-9.]*' gpurun_out/r5_b512.log | head -1)"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_details.json"))
print({k["name"]: (k["launches"], round(k["ms"]/max(1,k["launches"])*1e3,1)) for k in d.get("kernels",[])})
PY
