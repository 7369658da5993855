set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 300 ./persist_probe cold 1792 7168 > ../../gpurun_out/r4_persist_probe_cold.txt 2>&1; cd ../..
cat gpurun_out/r4_persist_probe_cold.txt
common="--batch 512 --right-context 13 --weights q8_0 --steps 20 --warmup 4 --regions 3 --no-b512 --no-f32-engine --no-host-pcm --no-cpu-baseline --no-buffered --no-extra-configs --checkpoint random --sync-steps"
timeout -k 10 600 python3 bench.py $common > gpurun_out/r4_b512_sync_persist.txt 2>&1; cp gpurun_out/bench_details.json gpurun_out/r4_b512_sync_persist_details.json
timeout -k 10 600 python3 bench.py $common --engine-option persistent_gemm=0 > gpurun_out/r4_b512_sync_nopersist.txt 2>&1; cp gpurun_out/bench_details.json gpurun_out/r4_b512_sync_nopersist_details.json
python3 - <<'PY'
import json
for t in ("persist","nopersist"):
    d=json.load(open(f"gpurun_out/r4_b512_sync_{t}_details.json"))
    print(t, d["line"]["ms_per_step"], [(k["name"],k["launches"],k["ms"]) for k in d["kernels"]][:12])
PY
for cfg in "burst 13 120" "burst 0 120" "realtime 0 20" "realtime 13 24"; do set -- $cfg
  timeout -k 10 600 python3 tests/server_load.py --streams 64 --seconds $3 --right-context $2 --mode $1 --warmup-seconds 6 --prewarm > gpurun_out/r4_server_load_$1_R$2.json 2> gpurun_out/r4_server_load_$1_R$2.err
  echo load $cfg rc $?
  grep prewarm gpurun_out/r4_server_load_$1_R$2.err
done
timeout -k 10 600 python3 bench.py --gpus 2 --share-device 0 --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline --no-extra-configs --no-profile-pass > gpurun_out/r4_two_ranks.txt 2> gpurun_out/r4_two_ranks.err
timeout -k 10 600 python3 bench.py --stream-offset 1 --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline --no-extra-configs --no-profile-pass > gpurun_out/r4_one_rank_stream1.txt 2> /dev/null
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline --no-extra-configs --no-profile-pass > gpurun_out/r4_one_rank_stream0.txt 2> /dev/null
echo done
