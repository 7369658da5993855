cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tests/micro/gemm_variant_identity.py "opt:dwconv_stream=0" "opt:resid_epilogue=0" > gpurun_out/r5_identity.log 2>&1; echo "identity rc=$?"; tail -4 gpurun_out/r5_identity.log
timeout -k 10 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_round5.py -m gpu -x -q -k "config3 or shipped_pipeline or 512" > gpurun_out/r5_cfg.log 2>&1; echo "cfg rc=$?"; tail -3 gpurun_out/r5_cfg.log
C64="--batch 64 --right-context 13 --weights q8_0 --no-grouped --no-cpu-baseline --no-extra-configs --no-buffered --no-host-pcm --no-f32-engine --no-b512 --steps 100 --warmup 8"
C512="--batch 512 --right-context 13 --weights q8_0 --no-grouped --no-cpu-baseline --no-extra-configs --no-buffered --no-host-pcm --no-f32-engine --no-b512 --steps 20 --warmup 4 --regions 3"
for o in "dwconv_stream=1" "dwconv_stream=0"; do
timeout -k 10 200 python bench.py $C64 --engine-option $o > gpurun_out/r5_ab64_$o.log 2>&1; echo "b64 $o rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_ab64_$o.log | head -1) sync $(grep -o '"synchronous_ms_per_step": [0-9.]*' gpurun_out/r5_ab64_$o.log | head -1)"
done
for o in "ablate=0" "dwconv_stream=0" "resid_epilogue=0" "ablate=64"; do
timeout -k 10 300 python bench.py $C512 --engine-option $o > gpurun_out/r5_ab512_$o.log 2>&1; echo "b512 $o rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_ab512_$o.log | head -1) sync $(grep -o '"synchronous_ms_per_step": [0-9.]*' gpurun_out/r5_ab512_$o.log | head -1)"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_details.json"))
print({k["name"]: round(k["ms"]/max(1,k["launches"])*1e3,1) for k in d.get("kernels",[]) if k["name"] in ("k_attention","k_dwconv","k_post","k_gemm_tiled")})
PY
done
