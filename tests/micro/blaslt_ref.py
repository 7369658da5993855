"""micro-benchmark: what the vendor GEMM library (hipBLASLt / rocBLAS through torch.matmul) takes for the five GEMM shapes of a
conformer layer at 64 streams x R = 13 (M = 896 rows, bf16) -- a reference point for k_gemm_roles / k_gemm_t64, whose epilogues
(bias, SiLU, GLU, split-K partials, K/V scatter) the library calls do not have.  Run under rocprofv3 --kernel-trace --stats for
per-kernel durations, or alone for event timings.
usage: python3 tests/micro/blaslt_ref.py [M]"""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 896
shapes = [("W1   (N=4096,K=1024)", 4096, 1024), ("W2   (N=1024,K=4096)", 1024, 4096), ("QKV  (N=3072,K=1024)", 3072, 1024),
          ("pw1  (N=2048,K=1024)", 2048, 1024), ("Wo/pw2 (N=1024,K=1024)", 1024, 1024)]
dev = torch.device("cuda:0")
for name, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(24)]      # 24 layers' worth: no weight reuse from cache
    for w in ws[:3]:
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        for w in ws:
            torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (reps * len(ws))
    print(f"{name}: {us:6.2f} us per GEMM back to back = {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
