set -x
mkdir -p gpurun_out
NASR_REPORT_DIR=gpurun_out/r4_reports timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite5.txt 2>&1
echo suite rc $?
tail -4 gpurun_out/r4_gpu_suite5.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_final3.json 2> gpurun_out/r4_bench_final3.err
echo bench rc $?
python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_final3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['configs']['b64_R13_q8_0']['ms_per_step'], d['configs']['b64_R13_q8_0']['step_mfma_frac'], d['configs']['b512_R13_q8_0'], d['configs']['b64_R13_diarize'], d['cpu_baseline']['value'], d['f32_engine']['b64_R13_ms_per_step'])
"
