mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_diar.py -x -q > gpurun_out/r6/t_fft.log 2>&1; echo "tests rc=$?"; tail -n 2 gpurun_out/r6/t_fft.log
bash tests/prof_diar.sh > gpurun_out/r6/diar_kernels_fft.txt 2>&1
grep "k_diar_logmel\|k_diar_frames\|k_vad_marblenet_h16" gpurun_out/r6/diar_kernels_fft.txt
grep "VAD on the bf16 MFMA\] VAD\|device-resident" gpurun_out/diar_bench.log
