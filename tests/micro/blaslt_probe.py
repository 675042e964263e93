"""What the library GEMM (hipBLASLt through torch.matmul) takes on the DiT block's four products at the bench's M = 6400 - a yardstick
for gemm256_k's tilings, not a product path (the library has no gated-residual / rotary epilogue)."""
import torch, sys
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("ff1", 2048, 1024), ("ff2", 1024, 2048)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(20): y = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n): y = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name}: M {M} N {N} K {K}  {us:.1f} us  {2.0*M*N*K/us*1e-6:.0f} TFLOP/s", flush=True)
