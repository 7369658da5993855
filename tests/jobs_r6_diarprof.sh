mkdir -p gpurun_out/r6
bash tests/prof_diar.sh > gpurun_out/r6/diar_kernels_final.txt 2>&1; echo "diar prof rc=$?"
cp $(ls -t gpurun_out/prof_diar/*/*kernel_stats.csv | head -1) gpurun_out/r6/diar_kernel_stats_final.csv
cp $(ls -t gpurun_out/prof_diar/*/*kernel_trace.csv | head -1) gpurun_out/r6/diar_kernel_trace_final.csv
grep "TitaNet\|VAD:" gpurun_out/diar_bench.log
