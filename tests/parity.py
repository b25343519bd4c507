"""Error measures and the tolerance ledger of the GPU parity tests.

`rel_err`  : max |a - b| / max |b|  (one number for a whole array: a wrong 32 x 32 tile among small entries can hide in it).
`tile_err` : the array is cut into 32 x 32 tiles (vectors: segments of 32, the MFMA / workgroup tile of every kernel
             here); per tile  ||a - b||_F / max(||b||_F, floor)  with floor = 1e-3 of the RMS tile norm of `b`;
             the worst tile is returned.  A tile that is garbage, zero or transposed reads ~1 whatever the other tiles
             hold; only tiles whose reference content is below 1e-3 of the typical tile are measured against the floor.
`observe`  : assert `err <= tol`.  Every fp32 tolerance of the suite goes through it: `tools/observed_errors.py --run`
             runs the suite in its own process with THIS function wrapped by a recorder, and prints observed-vs-bound for
             the whole suite from one GPU run; the bound written at each call site is <= 10x the value observed on
             MI355X (stated in the comment beside it).  Nothing in here reads the environment: no variable can turn the
             assertions of a test run off.
"""
import numpy as np


def _f64(a):
    if hasattr(a, "detach"):
        a = a.detach().double().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b):
    a, b = _f64(a).reshape(-1), _f64(b).reshape(-1)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def tile_err(a, b, tile=32):
    a, b = _f64(a), _f64(b)
    if a.shape != b.shape and a.size == b.size and np.squeeze(a).shape == np.squeeze(b).shape:
        b = b.reshape(a.shape)      # [M] against [1, M]: the same leaf with a unit axis
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.ndim == 0 or a.size == 1:
        return rel_err(a, b)
    if a.ndim == 1 or (a.ndim == 2 and 1 in a.shape):
        a, b = a.reshape(1, -1), b.reshape(1, -1)
    lead = int(np.prod(a.shape[:-2])) if a.ndim > 2 else 1
    a, b = a.reshape(lead, a.shape[-2], a.shape[-1]), b.reshape(lead, b.shape[-2], b.shape[-1])
    R, C = a.shape[1], a.shape[2]
    tr, tc = min(tile, R), min(tile, C)
    pr, pc = (-R) % tr, (-C) % tc
    pad = lambda x: np.pad(x, ((0, 0), (0, pr), (0, pc)))
    d2 = pad((a - b) ** 2).reshape(lead, (R + pr) // tr, tr, (C + pc) // tc, tc).sum(axis=(2, 4))
    b2 = pad(b ** 2).reshape(lead, (R + pr) // tr, tr, (C + pc) // tc, tc).sum(axis=(2, 4))
    nz = b2[b2 > 0]
    floor2 = 1e-6 * (nz.mean() if nz.size else 1.0)
    return float(np.sqrt((d2 / np.maximum(b2, floor2)).max()))


def prod_err(got, ref, absA, absB):
    """Componentwise forward error of a matrix product A B against its natural scale: max |got - ref| / (|A| |B|).
    Entries of the product that are small by CANCELLATION of large terms (rows of L^-1 against columns of K) carry the
    rounding of those terms; measured against the tile's own norm they read as large relative errors although the
    product is as accurate as fp32 allows.  A wrong tile still reads ~1.  Entries whose scale is below 1e-12 of the largest
    (RBF values past 7 lengthscales: fp32 underflow territory) are measured against that floor."""
    got, ref = _f64(got), _f64(ref)
    scale = _f64(absA) @ _f64(absB)
    return float((np.abs(got - ref) / np.maximum(scale, 1e-12 * scale.max())).max())


def observe(name, err, tol):
    err = float(err)
    assert np.isfinite(err), "%s: error is not finite" % name
    assert err <= tol, "%s: error %.3e exceeds the bound %.3e" % (name, err, tol)
    return err
