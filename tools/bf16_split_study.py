"""Accuracy of split-bf16 operands (fp32 accumulate) for the A = L^-1 K(z,x) contraction, emulated on the CPU.

bf16 keeps 8 significand bits.  x = x0 + x1 + x2 with x_k = bf16(residual): three terms carry 24 bits.  The MFMA would
accumulate the products in fp32.  Variants:  1 term (plain bf16), 2 terms x 3 products, 3 terms x 6 products.
Reference: the same product in float64.  Data: cfg-2-like Kmm (M = 512, spacing 0.5 lengthscales, jitter 1e-4).
"""
import numpy as np

def bf16(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF          # round to nearest even
    return ((u + r) & 0xFFFF0000).view(np.float32)

def split(x, terms):
    out, r = [], np.asarray(x, dtype=np.float32)
    for _ in range(terms):
        h = bf16(r)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out

def mm32(a, b):  # fp32 accumulate
    return a.astype(np.float32) @ b.astype(np.float32)

rng = np.random.RandomState(0)
M, n = 512, 2048
z = np.linspace(0, 256, M)[:, None]
x = rng.uniform(0, 256, (n, 1))
K = np.exp(-0.5 * (z - z.T) ** 2) + 1e-4 * np.eye(M)
L = np.linalg.cholesky(K)
W = np.linalg.inv(L)
Kzx = np.exp(-0.5 * (z - x.T) ** 2)
ref = W @ Kzx
scale = np.abs(ref).max()
print("max|A| = %.3f, max|W| = %.1f" % (scale, np.abs(W).max()))
print("fp32 operands, fp32 accumulate        : max abs err %.2e" % np.abs(mm32(W, Kzx) - ref).max())
for terms, pairs in ((1, [(0, 0)]), (2, [(0, 0), (0, 1), (1, 0)]), (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])):
    ws, ks = split(W, terms), split(Kzx, terms)
    acc = np.zeros_like(ref, dtype=np.float32)
    for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):  # small terms first
        acc = acc + mm32(ws[i], ks[j])
    print("bf16 x%d terms, %d products, fp32 acc : max abs err %.2e  (rel to max|A| %.1e)" % (terms, len(pairs), np.abs(acc - ref).max(), np.abs(acc - ref).max() / scale))
# effect on v = 1 - colsum(A^2) (what the ELBO sees)
v_ref = 1 - (ref ** 2).sum(0)
for terms, pairs in ((2, [(0, 0), (0, 1), (1, 0)]), (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])):
    ws, ks = split(W, terms), split(Kzx, terms)
    acc = sum(mm32(ws[i], ks[j]) for i, j in pairs)
    print("v = 1 - colsum(A^2), bf16 x%d : max abs err %.2e (fp32 operands: %.2e)" % (terms, np.abs(1 - (acc.astype(np.float64) ** 2).sum(0) - v_ref).max(), np.abs(1 - (mm32(W, Kzx).astype(np.float64) ** 2).sum(0) - v_ref).max()))
