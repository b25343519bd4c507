"""TEST INFRASTRUCTURE: evaluate a henbun_amd graph on the CPU with torch fp64.

Lets the `-m "not gpu"` suite check the host logic -- tracing, shape rules and
above all the graph-level autodiff (henbun_amd.graph.gradients) -- against
torch autograd of the oracle, without a GPU.  The product never imports this.
Each primitive is evaluated with the oracle's restatement of the same maths;
the fused `*_grad` primitives are evaluated by torch autograd of the oracle
forward (so the VJP *wiring* is what is under test here; the HIP kernels'
own arithmetic is tested on the GPU in tests/test_kernels_gpu.py).
"""
import math

import numpy as np
import torch

import henbun_oracle as O
from henbun_amd import graph as G

DT = torch.float64


def _ew(f, p, ins):
    a = ins[0]
    b = ins[1] if len(ins) > 1 else None
    un = {
        "NEG": lambda: -a, "EXP": lambda: a.exp(), "LOG": lambda: a.log(), "SQRT": lambda: a.sqrt(),
        "SQUARE": lambda: a * a, "ABS": lambda: a.abs(), "SIGN": lambda: a.sign(), "SIGMOID": lambda: torch.sigmoid(a),
        "RELU": lambda: torch.relu(a), "SOFTPLUS": lambda: torch.nn.functional.softplus(a), "TANH": lambda: torch.tanh(a),
        "RECIP": lambda: 1 / a, "RSQRT": lambda: a.rsqrt(), "STEP": lambda: (a > 0).to(DT),
        "AFFINE": lambda: p[0] * a + p[1], "CLIP": lambda: a.clamp(p[0], p[1]),
        "CLIPMASK": lambda: ((a >= p[0]) & (a <= p[1])).to(DT), "LGAMMA": lambda: torch.lgamma(a),
        "POWC": lambda: a ** p[0], "LOG1P": lambda: torch.log1p(a), "COPY": lambda: a.clone(),
        "DIGAMMA": lambda: torch.digamma(a),
        "ADD": lambda: a + b, "SUB": lambda: a - b, "MUL": lambda: a * b, "DIV": lambda: a / b,
        "MAX": lambda: torch.maximum(a, b), "MIN": lambda: torch.minimum(a, b), "POW": lambda: a ** b,
        "GT": lambda: (a > b).to(DT), "GE": lambda: (a >= b).to(DT), "LT": lambda: (a < b).to(DT),
        "LE": lambda: (a <= b).to(DT), "EQ": lambda: (a == b).to(DT),
        "SIGMOID_GRAD": lambda: b * a * (1 - a), "TANH_GRAD": lambda: b * (1 - a * a),
        "RELU_GRAD": lambda: b * (a > 0).to(DT), "SOFTPLUS_GRAD": lambda: b * torch.sigmoid(a),
        "CLIP_GRAD": lambda: b * ((a >= p[0]) & (a <= p[1])).to(DT),
        "WHERE": lambda: torch.where(a != 0, ins[1], ins[2]), "FMA": lambda: a * b + ins[2],
        "GAUSS_LOGPDF": lambda: O.gaussian(a, b, ins[2]),
    }
    if f == "GAUSS_LOGPDF_GRAD":
        x, mu, var, g = ins
        shp = torch.broadcast_shapes(x.shape, mu.shape, var.shape, g.shape)
        d = mu - x
        return [(g * d / var).expand(shp), (-g * d / var).expand(shp),
                (g * (-0.5 / var + 0.5 * d * d / (var * var))).expand(shp)]
    return [un[f]()]


def _strided(x, shape, strides, offset):
    return torch.as_strided(x.contiguous().reshape(-1), tuple(shape), tuple(strides), offset).clone()


def evaluate(outputs, leaf_values, noise=None):
    """leaf_values: {leaf Tensor: array}; noise: {noise-leaf or fused-op node id: array}.
    Returns {Tensor: torch tensor} for every tensor reachable from outputs."""
    vals = {}
    noise = noise or {}
    rng = np.random.RandomState(1234)
    for n in G.topo_order(outputs):
        ins = [vals[t] for t in n.inputs]
        op, at = n.op, n.attrs
        if op.startswith("leaf:"):
            t = n.outputs[0]
            kind = op[5:]
            if kind == "const":
                v = torch.as_tensor(np.asarray(at["value"]), dtype=DT)
            elif kind == "noise":
                v = torch.as_tensor(np.asarray(noise[t]) if t in noise else rng.randn(*t.shape), dtype=DT)
            else:
                lv = leaf_values[t]
                v = lv if isinstance(lv, torch.Tensor) else torch.as_tensor(np.asarray(lv), dtype=DT)
            outs = [v.reshape(t.shape)]
        elif op == "ew":
            outs = _ew(at["f"], at["p"], ins)
        elif op == "stop_gradient":
            outs = [ins[0]]
        elif op == "reduce":
            x = ins[0].reshape(at["K1"], at["R"], at["K2"])
            r = x.sum(1) if at["kind"] == "sum" else x.amax(1)
            outs = [r]
        elif op == "reshape":
            outs = [ins[0].reshape(at["shape"])]
        elif op == "gauss_ll":
            y, f_, var = ins[:3]
            s = ins[3].reshape(()) if len(ins) > 3 else torch.ones((), dtype=DT)
            v = var.reshape(())
            mu = f_.reshape(-1) * s
            dlt = y.reshape(-1) - mu
            dmu = dlt / v
            ll = (-0.5 * math.log(2 * math.pi) - 0.5 * torch.log(v) - 0.5 * dlt * dlt / v).sum()
            outs = [ll.reshape(1), dmu.reshape(f_.shape), (dmu * f_.reshape(-1)).sum().reshape(1),
                    (-0.5 / v + 0.5 * dlt * dlt / (v * v)).sum().reshape(1)]
        elif op == "bcast":
            outs = [torch.broadcast_to(ins[0], tuple(at["shape"])).clone()]
        elif op == "strided":
            outs = [_strided(ins[0], at["shape"], at["strides"], at["offset"])]
        elif op == "scatter_strided":
            out = torch.zeros(int(np.prod(at["xshape"])), dtype=DT)
            view = torch.as_strided(out, tuple(at["shape"]), tuple(at["strides"]), at["offset"])
            view.copy_(ins[0].reshape(at["shape"]))
            outs = [out]
        elif op == "concat":
            outs = [torch.cat(ins, dim=at["axis"])]
        elif op == "matmul":
            a, b = ins[0], ins[1]
            a = a.transpose(-1, -2) if at["ta"] else a
            b = b.transpose(-1, -2) if at["tb"] else b
            y = a @ b
            if at.get("actgrad"):
                yy = ins[2]
                d = {"sigmoid": yy * (1 - yy), "relu": (yy > 0).to(DT), "tanh": 1 - yy * yy}[at["actgrad"]]
                outs = [y * d]
            elif len(ins) > 2:
                y = y + ins[2].reshape(ins[2].shape[:-2] + (1, ins[2].shape[-1])) if ins[2].dim() >= 2 else y + ins[2]
            if not at.get("actgrad"):
                y = {"none": lambda v: v, "sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}[at["act"]](y)
                outs = [y]
        elif op == "matutil":
            x = ins[0]
            m = at["mode"]
            if m == 0:
                r, c = x.shape[-2], x.shape[-1]
                i = torch.arange(r)[:, None]
                j = torch.arange(c)[None, :]
                keep = ((at["lower"] < 0) | (i - j <= at["lower"])) & ((at["upper"] < 0) | (j - i <= at["upper"]))
                outs = [x * keep.to(DT)]
            elif m == 1:
                outs = [x + at["alpha"] * torch.eye(x.shape[-1], dtype=DT)]
            elif m == 2:
                outs = [torch.tril(x, -1) + 0.5 * torch.diag_embed(torch.diagonal(x, dim1=-2, dim2=-1))]
            elif m == 4:
                lo = torch.tril(x)
                outs = [0.5 * (lo + torch.tril(x, -1).transpose(-1, -2))]
            else:
                outs = [0.5 * (x + x.transpose(-1, -2))]
        elif op == "cholesky":
            outs = [torch.linalg.cholesky(ins[0])]
        elif op == "trinv":
            L = torch.tril(ins[0])
            eye = torch.eye(L.shape[-1], dtype=DT).expand(L.shape)
            outs = [torch.tril(torch.linalg.solve_triangular(L, eye, upper=False))]
        elif op in ("diag_sample_kl", "fullrank_sample_kl"):
            mu, s = ins[0], ins[1]
            if len(ins) > 2:
                u = ins[2]
            else:
                u = torch.as_tensor(noise[n.id] if n.id in noise else rng.randn(*mu.shape), dtype=DT).reshape(mu.shape)
            qs = "diagonal" if op == "diag_sample_kl" else "fullrank"
            x = O.sample_diag(mu, s, u) if qs == "diagonal" else O.sample_fullrank(mu, s, u)
            outs = [x, O.kl_normal(s, u, x, qs).reshape(1), u]
        elif op == "mlp2_sample_kl":
            # the amortised encoder as one op (graph.mlp2_sample_kl): evaluated op by op with the oracle's pieces
            y, w0, b0, w1, b1 = ins[:5]
            actf = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}[at["act"]]
            o = actf(y @ w0 + b0.reshape(1, -1)) @ w1 + b1.reshape(1, -1)
            L = o.shape[1] // 2
            mu, s = o[:, :L], o[:, L:]
            if len(ins) > 5:
                u = ins[5].reshape(mu.shape)
            else:
                u = torch.as_tensor(noise[n.id] if n.id in noise else rng.randn(*mu.shape), dtype=DT).reshape(mu.shape)
            x = O.sample_diag(mu, s, u)
            outs = [x, O.kl_normal(s, u, x, "diagonal").reshape(1), u, o]
        elif op == "mlp2_sample_kl_grad":
            y, w0, b0, w1, o, u, x = ins[:7]
            k = 7
            xbar, klbar = torch.zeros_like(x), torch.zeros(1, dtype=DT)
            if at["has_x"]:
                xbar = ins[k]
                k += 1
            if at["has_kl"]:
                klbar = ins[k]
            L = o.shape[1] // 2
            mb = xbar + klbar * x
            do = torch.cat([mb, mb * torch.exp(o[:, L:]) * u - klbar], dim=1)
            actf = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}[at["act"]]
            h = actf(y @ w0 + b0.reshape(1, -1))
            dact = {"sigmoid": h * (1 - h), "relu": (h > 0).to(DT), "tanh": 1 - h * h}[at["act"]]
            dh = (do @ w1.T) * dact
            outs = [y.T @ dh, dh.sum(0).reshape(b0.shape), h.T @ do, do.sum(0).reshape(n.outputs[3].shape)]
        elif op in ("diag_sample_kl_grad", "fullrank_sample_kl_grad"):
            s, u, x = ins[:3]
            k = 3
            xbar = torch.zeros_like(x)
            klbar = torch.zeros(1, dtype=DT)
            if at["has_x"]:
                xbar = ins[k]
                k += 1
            if at["has_kl"]:
                klbar = ins[k]
            mb = xbar + klbar * x
            if op == "diag_sample_kl_grad":
                outs = [mb, mb * torch.exp(s) * u - klbar]
            else:
                Sb = torch.tril(mb[..., :, None] * u[..., None, :]) - klbar * torch.diag_embed(
                    1.0 / torch.diagonal(s, dim1=-2, dim2=-1))
                outs = [mb, Sb]
        elif op == "gram":
            X, X2, ell = ins
            B = max(X.dim(), X2.dim())
            kind = at["kind"]
            if kind == "rbf":
                outs = [O.rbf_K(X, X2, ell)]
            elif kind == "csym_rbf":
                outs = [O.csym_rbf_K(X, X2, ell)]
            else:
                outs = [O.square_dist(X, X2, ell)]
        elif op == "gram_grad":
            X, X2, ell, g = ins
            Xr, X2r, lr = [t.clone().requires_grad_(True) for t in (X, X2, ell)]
            kind = at["kind"]
            K = O.rbf_K(Xr, X2r, lr) if kind == "rbf" else (O.csym_rbf_K(Xr, X2r, lr) if kind == "csym_rbf"
                                                             else O.square_dist(Xr, X2r, lr))
            gX, gX2, gl = torch.autograd.grad((K * g).sum(), [Xr, X2r, lr])
            outs = [gX + gX2, gl] if at.get("sym") else [gX, gX2, gl]
        elif op in ("sgp", "sgp_grad"):
            if op == "sgp":
                x, z, ell, L, W, u = ins[:6]
                eps = ins[6] if len(ins) > 6 else torch.as_tensor(
                    noise[n.id] if n.id in noise else rng.randn(*n.outputs[3].shape), dtype=DT)
                outs = list(_sgp_forward(x, z, ell, L, u, eps, at["mode"]))
            else:
                x, z, ell, W, u, eps, A, v, gf = ins
                L = torch.linalg.inv(W)
                xr, zr, lr, Lr, ur = [t.clone().requires_grad_(True) for t in (x, z, ell, L, u)]
                f = _sgp_forward(xr, zr, lr, Lr, ur, eps, at["mode"])[0]
                gL, gu, gz, gl, gx = torch.autograd.grad((f * gf).sum(), [Lr, ur, zr, lr, xr])
                outs = [torch.tril(gL), gu, gz, gl, gx]
        else:
            raise NotImplementedError("graph_oracle: op " + op)
        for t, v in zip(n.outputs, outs):
            vals[t] = v.reshape(t.shape)
    return vals


def _sgp_forward(x, z, ell, L, u, eps, mode):
    def one(x, z, ell, L, u, eps):
        A = torch.linalg.solve_triangular(torch.tril(L), O.rbf_K(z, x, ell), upper=False)
        mean = u @ A
        v = 1.0 - (A * A).sum(0)
        f = mean + torch.sqrt(torch.abs(v)) * eps if mode == "diagonal" else mean
        return f, A, v, eps

    if z.dim() == 2:
        return one(x, z, ell, L, u, eps)
    res = [one(x if x.dim() == 2 else x[e], z[e], ell[e], L[e], u[e], eps[e]) for e in range(z.shape[0])]
    return tuple(torch.stack([r[i] for r in res]) for i in range(4))
