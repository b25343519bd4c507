"""Time the GEMMs of the cfg-4 amortised encoder step (reference nn.py:31-32, 73-84) one by one: forward layers, input
gradients (activation-gradient epilogue) and weight + bias gradients.  HIP events, 100 launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H


def t(fn, iters=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


n = 32768
r = lambda *s: torch.randn(*s, device="cuda")
Y, W1, b1, W2, b2, Wd, bd = r(n, 64), r(64, 256), r(256), r(256, 32), r(32), r(16, 64), r(64)
h, o2, z, g64, g32, dh = torch.rand(n, 256, device="cuda"), r(n, 32), r(n, 16), r(n, 64), r(n, 32), r(n, 256)
out_h, out_o, out_d, out_z = torch.empty_like(h), torch.empty_like(o2), torch.empty_like(g64), torch.empty_like(z)
rows = [
    ("fwd  [n,64]x[64,256]+b sigmoid", lambda: H.matmul(Y, W1, bias=b1, act="sigmoid", out=out_h), 2 * n * 64 * 256, 4 * (n * 64 + n * 256)),
    ("fwd  [n,256]x[256,32]+b", lambda: H.matmul(h, W2, bias=b2, out=out_o), 2 * n * 256 * 32, 4 * (n * 256 + n * 32)),
    ("fwd  [n,16]x[16,64]+b", lambda: H.matmul(z, Wd, bias=bd, out=out_d), 2 * n * 16 * 64, 4 * (n * 16 + n * 64)),
    ("dx   [n,64]x[16,64]^T", lambda: H.matmul(g64, Wd, transB=True, out=out_z), 2 * n * 16 * 64, 4 * (n * 16 + n * 64)),
    ("dx   [n,32]x[256,32]^T * act'(h)", lambda: H.matmul(g32, W2, transB=True, act="sigmoid", actgrad=h, out=dh), 2 * n * 256 * 32, 4 * (n * 32 + 2 * n * 256)),
    ("dW   h^T g [256,32] + colsum", lambda: H.matmul_colsum(h, g32), 2 * n * 256 * 32, 4 * (n * 256 + n * 32)),
    ("dW   Y^T dh [64,256] + colsum", lambda: H.matmul_colsum(Y, dh), 2 * n * 64 * 256, 4 * (n * 64 + n * 256)),
    ("dW   z^T g [16,64] + colsum", lambda: H.matmul_colsum(z, g64), 2 * n * 16 * 64, 4 * (n * 16 + n * 64)),
]
rows += [
    ("x    [n,64]x[64,256] no epilogue", lambda: H.matmul(Y, W1, out=out_h), 2 * n * 64 * 256, 4 * (n * 64 + n * 256)),
    ("x    [n,64]x[64,128]+b sigmoid", lambda: H.matmul(Y, W1[:, :128].contiguous(), bias=b1[:128].contiguous(), act="sigmoid", out=out_h[:, :128].contiguous()), 2 * n * 64 * 128, 4 * (n * 64 + n * 128)),
]
rows += [
    ("ref  fill [n,256]", lambda: H.fill(out_h, 0.5), 0, 4 * n * 256),
    ("ref  sigmoid [n,256] -> [n,256]", lambda: H.ewise("SIGMOID", [h], out=out_h), 0, 8 * n * 256),
    ("ref  copy [n,64] -> [n,64]", lambda: H.ewise("COPY", [Y], out=out_d), 0, 8 * n * 64),
]
sel = [int(a) for a in sys.argv[1:]]
if sel:
    rows = [rows[i] for i in sel]
tot = 0.0
for lab, fn, fl, by in rows:
    us = t(fn)
    tot += us
    print("%-36s %7.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (lab, us, fl / us * 1e-6, by / us * 1e-6), flush=True)
print("sum %.1f us" % tot)
