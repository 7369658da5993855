set -x
mkdir -p gpurun_out
AB_SYNC=0 timeout -k 10 600 bash tests/micro/ab_b64.sh "GPU_MAX_HW_QUEUES=5" "GPU_MAX_HW_QUEUES=8" > gpurun_out/r4_ab_b64_hwq.txt 2>&1; cat gpurun_out/r4_ab_b64_hwq.txt
GPU_MAX_HW_QUEUES=5 NASR_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 100 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline --no-extra-configs --no-profile-pass > gpurun_out/r4_hwq5_b1.txt 2>&1
python3 -c "
import json
l=[x for x in open('gpurun_out/r4_hwq5_b1.txt').read().splitlines() if x.startswith('{')][-1]
d=json.loads(l); print('b1 hwq5', d['value'], d['ms_per_step'], d.get('per_rank'))"
AB_SYNC=0 AB_BATCH=16 timeout -k 10 400 bash tests/micro/ab_b64.sh "GPU_MAX_HW_QUEUES=5" > gpurun_out/r4_ab_b16_hwq.txt 2>&1; cat gpurun_out/r4_ab_b16_hwq.txt
