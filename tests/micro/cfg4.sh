# configs[4] alone (64 streams x R = 13 from Q8_0 tensors + the diarization side-car), with variants of how the queues are shared
# usage: bash tests/micro/cfg4.sh [tag]
set -x
mkdir -p gpurun_out/r6
COMMON="--batch 64 --right-context 13 --weights q8_0 --steps 40 --warmup 5 --diarize --no-extra-configs --no-cpu-baseline --no-f32-engine --no-buffered --no-grouped --no-host-pcm --no-profile-pass --no-b512"
for v in "0 3"; do
  set -- $v; s=$1; export NASR_DIAR_ASR_LANES=$2
  NASR_DIAR_SPLIT=$s python bench.py $COMMON > gpurun_out/r6/cfg4_split$s.json 2> gpurun_out/r6/cfg4_split$s.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r6/cfg4_split$s.json") if x.startswith("{")][-1]
d=json.loads(l)
print("split=$s lanes=$NASR_DIAR_ASR_LANES", "asr ms_per_step", d.get("ms_per_step"), "diarization", d.get("diarization"))
PY
done
