mkdir -p gpurun_out/r6/server
for c in 16 32 64; do
  timeout -k 10 400 python tests/server_load.py --streams 64 --seconds 240 --right-context 13 --mode burst --client native --conns $c --warmup-seconds 20 --backlog-chunks 4 > gpurun_out/r6/server/burst_R13_64_k4_c$c.json 2> gpurun_out/r6/server/burst_R13_64_k4_c$c.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6/server/burst_R13_64_k4_c$c.json") if l.startswith("{")][-1])
print("conns=$c", d["aggregate_rtfx"], d["wall_seconds"], d["transcripts_correct"], d["server"]["engine_calls"], d["server"]["streams_per_call_mean"])
PY
done
timeout -k 10 400 python tests/server_load.py --streams 64 --seconds 240 --right-context 13 --mode burst --client native --conns 32 --warmup-seconds 20 --backlog-chunks 1 > gpurun_out/r6/server/burst_R13_64_k1_c32.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6/server/burst_R13_64_k1_c32.json") if l.startswith("{")][-1])
print("k=1 conns=32", d["aggregate_rtfx"], d["wall_seconds"], d["transcripts_correct"], d["server"]["engine_calls"], d["server"]["streams_per_call_mean"])
PY
