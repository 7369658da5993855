"""Measured-achievable peaks of the box (SURVEY.md section 8d asks for them next to the vendor figures):
HBM stream copy and a large bf16 GEMM through the vendor library."""
import time
import torch

dev = "cuda"
n = 1 << 30                                  # 1 GiB each way
a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"stream copy 1 GiB -> 1 GiB: {2 * n / dt / 1e12:.2f} TB/s (read + write)")
c = torch.empty(n // 4, dtype=torch.float32, device=dev).normal_()
for _ in range(3):
    s = c.sum()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    s = c.sum()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"read-only reduction of 1 GiB: {n / dt / 1e12:.2f} TB/s")
for N in (4096, 8192):
    x = torch.randn(N, N, device=dev, dtype=torch.bfloat16)
    y = torch.randn(N, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        z = x @ y
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        z = x @ y
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"bf16 GEMM {N}^3 (hipBLASLt, random data): {2 * N ** 3 / dt / 1e12:.0f} TFLOP/s")
