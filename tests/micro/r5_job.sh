cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tests/micro/gemm_variant_identity.py "opt:post_rows=0" > gpurun_out/r5_identity.log 2>&1; echo "identity rc=$?"; tail -2 gpurun_out/r5_identity.log
C512="--batch 512 --right-context 13 --weights q8_0 --no-grouped --no-cpu-baseline --no-extra-configs --no-buffered --no-host-pcm --no-f32-engine --no-b512 --steps 20 --warmup 4 --regions 3"
for o in "post_rows=1" "post_rows=0"; do
timeout -k 10 300 python bench.py $C512 --engine-option $o > gpurun_out/r5_b512_$o.log 2>&1; echo "b512 $o rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_b512_$o.log | head -1) sync $(grep -o '"synchronous_ms_per_step": [0-9.]*' gpurun_out/r5_b512_$o.log | head -1)"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_details.json"))
print({k["name"]: (k["launches"], round(k["ms"]/max(1,k["launches"])*1e3,1)) for k in d.get("kernels",[]) if k["name"] in ("k_post","k_attention","k_dwconv","k_gemm_tiled")})
PY
done
