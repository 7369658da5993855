set -x
mkdir -p gpurun_out
for M in 896 3584 7168 15360; do echo "M = $M" >> gpurun_out/r4_blaslt_ref.txt; timeout -k 10 200 python3 tests/micro/blaslt_ref.py $M >> gpurun_out/r4_blaslt_ref.txt 2>&1; done
cat gpurun_out/r4_blaslt_ref.txt
timeout -k 10 100 python3 __graft_entry__.py smoke > gpurun_out/r4_smoke.txt 2>&1
echo smoke rc $?
tail -3 gpurun_out/r4_smoke.txt
NASR_REPORT_DIR=gpurun_out/r4_reports timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite2.txt 2>&1
echo suite rc $?
tail -5 gpurun_out/r4_gpu_suite2.txt
