"""hb_cholesky_inverse_f32 as one persistent launch (csrc/chol_persist.cuh) against numpy fp64 and against the
launch-chain form (hb_debug_set chol_persist 0): accuracy, info, repeatability, time per call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from henbun_amd import hip_ops as H  # noqa: E402


def spd(rng, B, M, jit):
    X = np.sort(rng.uniform(0, 0.4 * M, (B, M, 1)), axis=1)
    K = np.exp(-0.5 * (X - np.transpose(X, (0, 2, 1))) ** 2)
    return K + jit * np.eye(M)


def tile_err(a, ref, t=32):
    a, ref = np.atleast_3d(a), np.atleast_3d(ref)
    worst = 0.0
    nrm = np.sqrt(np.mean(ref.astype(np.float64) ** 2)) * t
    for b in range(a.shape[0]):
        for i in range(0, a.shape[1], t):
            for j in range(0, a.shape[2], t):
                d = np.linalg.norm(a[b, i:i + t, j:j + t].astype(np.float64) - ref[b, i:i + t, j:j + t])
                r = max(np.linalg.norm(ref[b, i:i + t, j:j + t]), 1e-3 * nrm)
                worst = max(worst, d / r)
    return worst


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    rng = np.random.RandomState(0)
    ok = True
    cases = [(1, 64), (1, 128), (2, 256), (1, 512), (3, 512), (8, 512), (1, 1024)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) for v in c.split("x")) for c in sys.argv[1:]]
    for B, M in cases:
        A = spd(rng, B, M, 1e-2)
        ref = np.linalg.cholesky(A)
        refW = np.linalg.inv(ref)
        a = torch.as_tensor(A, dtype=torch.float32).cuda().contiguous()
        H.debug_set("chol_persist", 1)
        L, W, info = H.cholesky_inverse(a)
        torch.cuda.synchronize()
        Lh, Wh = L.cpu().numpy(), W.cpu().numpy()
        eL, eW = tile_err(Lh, ref), tile_err(Wh, refW)
        wl = np.abs(Wh.astype(np.float64) @ Lh.astype(np.float64) - np.eye(M)).max()
        up = bool(np.all(np.triu(Lh, 1) == 0) and np.all(np.triu(Wh, 1) == 0))
        # repeat: the sync words must be clean again, results bit-identical
        L2, W2, info2 = H.cholesky_inverse(a)
        torch.cuda.synchronize()
        same = bool(torch.equal(L, L2) and torch.equal(W, W2))
        H.debug_set("chol_persist", 0)
        L0, W0, info0 = H.cholesky_inverse(a)
        torch.cuda.synchronize()
        e0L, e0W = tile_err(L0.cpu().numpy(), ref), tile_err(W0.cpu().numpy(), refW)
        t0 = timeit(lambda: H.cholesky_inverse(a, out=L0, inv=W0, info=info0))
        H.debug_set("chol_persist", 1)
        t1 = timeit(lambda: H.cholesky_inverse(a, out=L, inv=W, info=info))
        good = info.cpu().tolist() == [0] * B and eL < 5e-5 and eW < 5e-4 and wl < 5e-5 and up and same
        ok &= good
        print("B=%d M=%4d  persistent: L %.2e W %.2e WL-I %.2e upper0 %s repeat-identical %s info %s | chain: L %.2e W %.2e | "
              "us/call persistent %.1f chain %.1f  %s" % (B, M, eL, eW, wl, up, same, info.cpu().tolist(), e0L, e0W, t1, t0,
                                                         "ok" if good else "BAD"), flush=True)
    # failure reporting: a non-positive leading minor
    for M, col in ((128, 70), (512, 300), (512, 5)):
        A = spd(rng, 2, M, 1e-2)
        A[1, col, col] = -1.0
        a = torch.as_tensor(A, dtype=torch.float32).cuda().contiguous()
        _, _, info = H.cholesky_inverse(a)
        torch.cuda.synchronize()
        got = info.cpu().tolist()
        good = got == [0, col + 1]
        ok &= good
        print("failure M=%d col=%d: info %s %s" % (M, col, got, "ok" if good else "BAD"), flush=True)
        # and the workspace is usable again afterwards
        A2 = spd(rng, 2, M, 1e-2)
        _, _, info = H.cholesky_inverse(torch.as_tensor(A2, dtype=torch.float32).cuda().contiguous())
        torch.cuda.synchronize()
        ok &= info.cpu().tolist() == [0, 0]
    print("ALL OK" if ok else "FAILURES", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    t = time.time()
    rc = main()
    print("%.1f s" % (time.time() - t))
    sys.exit(rc)
