#!/usr/bin/env python3
"""Yardstick: vendor-library fp32 GEMM / conv rates on the same GPU for the D3 shape (not part of the product)."""
import torch, torch.nn.functional as F
torch.backends.cuda.matmul.allow_tf32 = False
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (M, N, K) in [(17672, 512, 4096), (18432, 256, 2048), (73728, 128, 1024), (4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda")
    t = timeit(lambda: a @ b)
    print("sgemm %6dx%5dx%5d %8.1f us %6.1f TF" % (M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))
x = torch.randn(8, 256, 48, 48, device="cuda"); w = torch.randn(512, 256, 4, 4, device="cuda")
for cl in (False, True):
    xx = x.contiguous(memory_format=torch.channels_last) if cl else x
    ww = w.contiguous(memory_format=torch.channels_last) if cl else w
    t = timeit(lambda: F.conv2d(xx, ww, None, 1, 1))
    print("miopen conv D3 fwd channels_last=%s %8.1f us %6.1f TF" % (cl, t * 1e6, 74.12e9 / t / 1e12))
