"""Does the 32 KB row stride of A / Kbar (n = 8192 floats) hurt the Lbar = Kbar A^T contraction?  Same GEMM with padded rows."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from henbun_amd import _lib, hip_ops as H
from ctypes import c_void_p
M, n = 512, 8192
lib = _lib.lib()
ws = torch.empty(32 * M * M, dtype=torch.float32, device="cuda")
C = torch.empty(M, M, dtype=torch.float32, device="cuda")
for pad in (0, 16, 32, 64, 128, 256, 1024):
    ld = n + pad
    A = torch.randn(M, ld, dtype=torch.float32, device="cuda")
    B = torch.randn(M, ld, dtype=torch.float32, device="cuda")
    for flags, name in ((H.MM_TRIL_OUT, "tril"), (0, "full")):
        def run():
            lib.call("hb_matmul_f32", c_void_p(A.data_ptr()), c_void_p(B.data_ptr()), c_void_p(C.data_ptr()), 1, M, M, n, ld, ld, M,
                     0, 0, 0, 0, 1, -1.0, 0.0, None, 0, 0, flags, c_void_p(ws.data_ptr()), ws.numel(), H.stream())
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        print("ld = n + %4d  %s: %.1f us (matmul + split-K finish)" % (pad, name, e0.elapsed_time(e1) * 1e3 / 50), flush=True)

print("--- operands stored [n, M] (contraction over rows; loads contiguous along m) ---")
At = torch.randn(n, M, dtype=torch.float32, device="cuda")
Bt = torch.randn(n, M, dtype=torch.float32, device="cuda")
for flags, name in ((H.MM_TRIL_OUT, "tril"), (0, "full")):
    def run():
        lib.call("hb_matmul_f32", c_void_p(At.data_ptr()), c_void_p(Bt.data_ptr()), c_void_p(C.data_ptr()), 1, M, M, n, M, M, M,
                 0, 0, 0, 1, 0, -1.0, 0.0, None, 0, 0, flags, c_void_p(ws.data_ptr()), ws.numel(), H.stream())
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    print("[n,M] layout %s: %.1f us (matmul + split-K finish)" % (name, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
