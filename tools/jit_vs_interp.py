"""Per-op comparison of the compiled (hiprtc) and interpreted forms of a one-instruction elementwise program."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import henbun_amd as hb
from henbun_amd import hip_ops as H

def mode(m):
    cfg = hb.settings.get_settings(); cfg.runtime.ewise = m
    return hb.settings.temp_settings(cfg)

rng = np.random.RandomState(0)
n = 4096
for dt in (torch.float32, torch.float64):
    a = torch.as_tensor(np.abs(rng.randn(n)) + 0.3, dtype=dt).cuda()
    b = torch.as_tensor(rng.randn(n), dtype=dt).cuda()
    c = torch.as_tensor(np.abs(rng.randn(n)) + 0.5, dtype=dt).cuda()
    for name, opc in sorted(H.EW.items(), key=lambda kv: kv[1]):
        if name == "GAUSS_LOGPDF_GRAD":
            code, params, nout = [[opc, 3, 0, 1, 2]], [[1.0, 0.0]], 3
            oregs = [3, 4, 5]
        else:
            code, params, nout = [[opc, 3, 0, 1, 2]], [[1.5, 2.5]], 1
            oregs = [3]
        outs = []
        for m in ("jit", "interpret"):
            o = [torch.empty(n, dtype=dt, device="cuda") for _ in range(nout)]
            with mode(m):
                H.EwiseProgram(code, params, [a, b, c], [[1], [1], [1]], o, oregs, [[1]] * nout, [n]).launch()
            torch.cuda.synchronize()
            outs.append(o)
        worst = 0.0
        for x, y in zip(*outs):
            x, y = torch.nan_to_num(x, nan=7.0), torch.nan_to_num(y, nan=7.0)
            d = ((x - y).abs() / (y.abs() + 1e-30)).max().item()
            worst = max(worst, d)
        if worst != 0.0:
            print("%s %-18s max relative difference %.3e" % (str(dt)[6:], name, worst))
print("done")
