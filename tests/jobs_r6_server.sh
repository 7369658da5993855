set -x
mkdir -p gpurun_out/r6/server
for k in 4; do
  timeout -k 10 400 python tests/server_load.py --streams 64 --seconds 240 --right-context 13 --mode burst --client native --conns 8 --warmup-seconds 20 --backlog-chunks $k > gpurun_out/r6/server/burst_R13_64_k$k.json 2> gpurun_out/r6/server/burst_R13_64_k$k.err; echo "k=$k rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6/server/burst_R13_64_k$k.json") if l.startswith("{")][-1])
print("k=$k", {x: d.get(x) for x in ("rtfx","streams","correct","transcripts_correct","wall_s")}, d.get("server"))
PY
done
