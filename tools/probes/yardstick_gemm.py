"""Measuring stick, NOT part of the product path: what a plain large bf16 GEMM sustains on this box (torch.matmul ->
hipBLASLt), on random and on all-zero operands, for a few seconds each so that rocm-smi can be sampled beside it.
VERDICT r3 item 4: is 0.26-0.33 of the 2.5 PFLOP/s bf16 MFMA peak the board (power / clock) or the kernels?

    python tools/probes/yardstick_gemm.py <random|zeros> <seconds> [n]
"""
import sys
import time

import torch

kind, secs = sys.argv[1], float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = torch.device("cuda:0")
if kind == "zeros":
    a = torch.zeros(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(n, n, device=dev, dtype=torch.bfloat16)
else:
    a = torch.randn(n, n, device=dev).to(torch.bfloat16)
    b = torch.randn(n, n, device=dev).to(torch.bfloat16)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    torch.matmul(a, b, out=c)
torch.cuda.synchronize()
t0 = time.perf_counter()
it = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    it += 10
dt = (time.perf_counter() - t0) / it
print("yardstick bf16 GEMM %dx%dx%d %s operands: %.3f ms per call, %.1f TFLOP/s = %.3f of 2500" %
      (n, n, n, kind, dt * 1e3, 2.0 * n ** 3 / dt / 1e12, 2.0 * n ** 3 / dt / 1e12 / 2500.0), flush=True)
