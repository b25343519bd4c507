"""Early-start forward (csrc/sgp.hip: chol_sgp_fwd_kernel) against the two-launch sequence it replaces:

    python tools/early_fwd_check.py [M n E]

 * results: L, W, images bit-identical; f, v, the fragment-major A and the head's outputs equal up to the three-way
   summation of the last two row tiles; info; a non-PD matrix; repeated calls on one workspace (sync words left zero)
 * (side jobs riding on the same launch -- minibatch gather, sample of q(u) -- are exercised through the model: tools/ab_step.py,
   tests/test_model_gpu.py)
 * timing: factorisation + forward as two launches / as one
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from henbun_amd import hip_ops as H  # noqa: E402

M, n, E = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 8192, 1)
dev = "cuda"
rng = np.random.RandomState(0)
lead = (E,) if E > 1 else ()
z = torch.tensor(np.sort(rng.uniform(0, M / 2, lead + (M, 1)), axis=-2), dtype=torch.float32, device=dev)
x = torch.tensor(rng.uniform(0, M / 2, (n, 1)), dtype=torch.float32, device=dev)
ell = torch.ones(lead + (1,), dtype=torch.float32, device=dev)
u = torch.tensor(rng.randn(*(lead + (1, M))), dtype=torch.float32, device=dev)
eps = torch.tensor(rng.randn(*(lead + (n,))), dtype=torch.float32, device=dev)
y = torch.tensor(rng.randn(*(lead + (n,))), dtype=torch.float32, device=dev)
var = torch.tensor([0.3], dtype=torch.float32, device=dev)
K = torch.exp(-0.5 * (z - z.transpose(-1, -2)) ** 2) + 1e-3 * torch.eye(M, device=dev)


def run(early, Kmat=K, uu=u):
    B = E
    L, W = torch.empty_like(Kmat), torch.empty_like(Kmat)
    frag = torch.empty(2 * B * M * M, dtype=torch.float32, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    prec = H.PREC_NATIVE
    afrag = torch.empty(H.sgp_frag_elems(E, n, M, prec), dtype=torch.float32, device=dev)
    units = H.sgp_head_units(x, z, uu, prec, True, False, None)
    head = dict(y=y, var=var, scale=None, post=0.5, dmu=torch.empty_like(y), fbar=torch.empty_like(y),
                part=torch.empty(3 * units, dtype=torch.float32, device=dev), units=units)
    out = (torch.empty(lead + (1, n), device=dev), torch.empty(lead + (M, n), device=dev), torch.empty(lead + (n,), device=dev),
           torch.empty(lead + (n,), device=dev))
    fwd = lambda: H.sgp_fwd(x, z, ell, W, uu, eps_in=eps, mode=1, out=out, wfrag=frag, prec=prec, a_frag=afrag, skip_a=True, head=head)
    if early:
        assert H.sgp_rider_supported(x, z, uu, prec, True, False, None)
        H.sgp_rider_begin()
        fwd()
        assert H.sgp_rider_pending() == 1
        H.cholesky_inverse(Kmat, out=L, inv=W, info=info, frag=frag)
        assert H.sgp_rider_pending() == 0
        H.sgp_rider_flush()
    else:
        H.cholesky_inverse(Kmat, out=L, inv=W, info=info, frag=frag)
        fwd()
    torch.cuda.synchronize()
    return dict(L=L, W=W, frag=frag, info=info, f=out[0], v=out[2], afrag=afrag, dmu=head["dmu"], fbar=head["fbar"], part=head["part"])


ref = run(False)
got = run(True)
print("M %d  n %d  E %d   info %s / %s" % (M, n, E, ref["info"].tolist(), got["info"].tolist()))
for k in ("L", "W", "frag"):
    print("  %-6s bit-identical: %s" % (k, torch.equal(ref[k], got[k])))
for k in ("f", "v", "afrag", "dmu", "fbar", "part"):
    a, b = ref[k].double(), got[k].double()
    print("  %-6s max |diff| %.3e  (scale %.3e)  identical elements %.4f" % (k, (a - b).abs().max().item(), a.abs().max().item(),
                                                                        (a == b).double().mean().item()))
again = run(True)
print("  second early call on the same workspace: f identical to the first: %s" % torch.equal(again["f"], got["f"]))
for _ in range(20):
    again = run(True)
print("  22nd: %s" % torch.equal(again["f"], got["f"]))

# a matrix that is not positive definite: info > 0 from both forms, no hang
Kbad = K.clone()
Kbad[..., 100, 100] = -1.0
rb, gb = run(False, Kbad), run(True, Kbad)
print("  non-PD: info %s / %s" % (rb["info"].tolist(), gb["info"].tolist()))
ok = run(True)
print("  after the failure, a good matrix again: info %s, f identical %s" % (ok["info"].tolist(), torch.equal(ok["f"], got["f"])))


def timeit(fn, it=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def make(early):
    B = E
    L, W = torch.empty_like(K), torch.empty_like(K)
    frag = torch.empty(2 * B * M * M, dtype=torch.float32, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    prec = H.PREC_NATIVE
    afrag = torch.empty(H.sgp_frag_elems(E, n, M, prec), dtype=torch.float32, device=dev)
    units = H.sgp_head_units(x, z, u, prec, True, False, None)
    head = dict(y=y, var=var, scale=None, post=0.5, dmu=torch.empty_like(y), fbar=torch.empty_like(y),
                part=torch.empty(3 * units, dtype=torch.float32, device=dev), units=units)
    out = (torch.empty(lead + (1, n), device=dev), torch.empty(lead + (M, n), device=dev), torch.empty(lead + (n,), device=dev),
           torch.empty(lead + (n,), device=dev))
    fwd = lambda: H.sgp_fwd(x, z, ell, W, u, eps_in=eps, mode=1, out=out, wfrag=frag, prec=prec, a_frag=afrag, skip_a=True, head=head)

    def step():
        if early:
            H.sgp_rider_begin()
            fwd()
            H.cholesky_inverse(K, out=L, inv=W, info=info, frag=frag)
        else:
            H.cholesky_inverse(K, out=L, inv=W, info=info, frag=frag)
            fwd()
    return step


s0, s1 = make(False), make(True)
for rnd in range(3):
    print("  two launches %.1f us    one launch %.1f us" % (timeit(s0), timeit(s1)))
for naps in (4, 16):
    H.debug_set("sgp_early_poll_naps", naps)
    print("  (diagnostic) %d naps of 512 cycles between polls: %.1f us" % (naps, timeit(s1)))
    H.debug_set("sgp_early_diag", 2)
    print("  (diagnostic)      ... and strips skip their MFMA work: %.1f us" % timeit(s1))
    H.debug_clear()
H.debug_set("sgp_early_diag", 1)
print("  (diagnostic) strips leave at once: %.1f us" % timeit(s1))
H.debug_set("sgp_early_diag", 2)
print("  (diagnostic) strips wait but skip their MFMA work: %.1f us" % timeit(s1))
H.debug_set("sgp_early_diag", 4)
print("  (diagnostic) the fused launch without strip workgroups: %.1f us" % timeit(s1))
H.debug_clear()
H.debug_set("chol_persist", 1)
L_, W_ = torch.empty_like(K), torch.empty_like(K)
fr_ = torch.empty(2 * E * M * M, dtype=torch.float32, device=dev)
print("  factorisation alone: %.1f us" % timeit(lambda: H.cholesky_inverse(K, out=L_, inv=W_, frag=fr_)))
