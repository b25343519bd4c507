"""HBM GB/s of the bandwidth-bound kernels at the sizes where they ARE bandwidth bound:
full-rank sampler at M = 1024 (dense 4.2 MB and tri-packed 2.1 MB q_sqrt), encoder-fed diagonal sampler and the
minibatch row gather at cfg 4 (n = 32768, L = 16 / 64 columns), Adam at cfg 3 (1.05 M parameters).
Algorithmic bytes / HIP-event time per launch (200 launches) against the 8 TB/s HBM3E peak."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H

def t(fn, iters=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def row(name, us, byts):
    print("%-58s %8.2f us  %8.1f MB  %7.0f GB/s  %.3f of 8 TB/s" % (name, us, byts / 1e6, byts / us * 1e-3, byts / us * 1e-3 / 8000.0), flush=True)

f32 = torch.float32
dev = "cuda"
M = 1024
mu, u = torch.randn(1, M, device=dev), torch.randn(1, M, device=dev)
S = torch.randn(1, M, M, device=dev) * 0.01 + 0.1 * torch.eye(M, device=dev)
Sp = H.tri_to_vec(S)
out = (torch.empty_like(mu), torch.empty(1, device=dev), torch.empty_like(mu))
row("full-rank sampler + MC-KL, M=1024, dense q_sqrt", t(lambda: H.fullrank_sample_kl_fwd(mu, S, u_in=u, out=out)), (M * M + 3 * M) * 4)
row("full-rank sampler + MC-KL, M=1024, tri-packed q_sqrt", t(lambda: H.fullrank_sample_kl_fwd(mu, Sp, u_in=u, out=out, packed=True)), (M * (M + 1) // 2 + 3 * M) * 4)
xb, kb = torch.randn(1, M, device=dev), torch.ones(1, device=dev)
o2 = (torch.empty_like(mu), torch.empty_like(S))
row("full-rank sampler VJP, M=1024, dense", t(lambda: H.fullrank_sample_kl_bwd(S, u, out[0], xb, kb, out=o2)), (2 * M * M + 4 * M) * 4)
o3 = (torch.empty_like(mu), torch.empty_like(Sp))
row("full-rank sampler VJP, M=1024, tri-packed", t(lambda: H.fullrank_sample_kl_bwd(Sp, u, out[0], xb, kb, out=o3, packed=True)), (M * (M + 1) + 4 * M) * 4)
n, L = 32768, 16
mu2, s2 = torch.randn(n, L, device=dev), torch.randn(n, L, device=dev) * 0.1
rng = H.Rng(0)
o4 = (torch.empty_like(mu2), torch.empty(1, device=dev), torch.empty_like(mu2))
row("diag sampler + MC-KL, n x L = 32768 x 16, in-kernel noise", t(lambda: H.diag_sample_kl_fwd(mu2, s2, rng=rng, out=o4)), 4 * n * L * 4)
Y = torch.randn(400000, 64, device=dev)
idx = torch.randint(0, 400000, (n,), device=dev)
og = torch.empty(n, 64, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
row("minibatch row gather, 32768 rows x 64 columns", t(lambda: H.gather_rows(Y, idx, None, out=og, err=err)), (2 * n * 64 * 4 + n * 8))
P = 1024 * 1024 + 1024 + 1024 + 3
th, g, m, v = (torch.randn(P, device=dev) for _ in range(4))
v.abs_()
tt = torch.zeros(1, dtype=torch.int64, device=dev)
row("Adam (TF-1 rule), 1.05 M parameters (cfg 3)", t(lambda: H.adam_step(th, g, m, v, tt)), 7 * P * 4)
P2 = 512 * 1025 + 1024 * 2 + 3
row("Adam, cfg 3 with the tri-packed q_sqrt (0.53 M parameters)", t(lambda: H.adam_step(th[:P2], g[:P2], m[:P2], v[:P2], tt)), 7 * P2 * 4)
