"""What the vendor library (hipBLASLt through torch.matmul) reaches on the large-M GEMM shapes of the layer --
a yardstick for k_gemm_tiled2, not a dependency of the engine."""
import torch, time
torch.manual_seed(0)
dev = "cuda"
for (M, N, K) in [(896, 4096, 1024), (896, 1024, 4096), (896, 3072, 1024), (896, 1024, 1024), (896, 2048, 1024)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        c = a @ w.t()
    torch.cuda.synchronize()
    n = 200
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            c = a @ w.t()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt * 1e6:.2f} us  {2 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
