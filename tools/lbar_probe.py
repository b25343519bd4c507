"""Lbar = -tril(Kbar A^T) (512x512, contraction 8192) under forced tile / split settings (HB_MM_FORCE_BT / HB_MM_FORCE_S)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for shape, cfgs in (("8 65536", ((0, 0), (64, 32), (128, 16))),):
        for pad in (0, 32, 64, 256, 1056):
            for bt, s in cfgs:
                env = dict(os.environ, HB_MM_FORCE_BT=str(bt), HB_MM_FORCE_S=str(s)) if bt else dict(os.environ)
                env["LBAR_PAD"] = str(pad)
                subprocess.run([sys.executable, __file__, "child"] + shape.split(), env=env, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from henbun_amd import _lib, hip_ops as H
from ctypes import c_void_p
M = 512
E, n = int(sys.argv[2]), int(sys.argv[3])
lib = _lib.lib()
ws = torch.empty(64 * M * M * E, dtype=torch.float32, device="cuda")
C = torch.empty(E, M, M, dtype=torch.float32, device="cuda")
pad = int(os.environ.get("LBAR_PAD", "0"))
ld = n + pad
A = torch.randn(E, M, ld, dtype=torch.float32, device="cuda")
B = torch.randn(E, M, ld, dtype=torch.float32, device="cuda")
def run():
    lib.call("hb_matmul_f32", c_void_p(A.data_ptr()), c_void_p(B.data_ptr()), c_void_p(C.data_ptr()), E, M, M, n, ld, ld, M,
             M * ld, M * ld, M * M, 0, 1, -1.0, 0.0, None, 0, 0, H.MM_TRIL_OUT, c_void_p(ws.data_ptr()), ws.numel(), H.stream())
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        run()
    g = H.CapturedGraph()
    g.begin()
    for _ in range(10):
        run()
    g.end()
    g.launch(); st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); g.launch(); g.launch(); e1.record(st); st.synchronize()
print("pad=%d " % pad + "E=%d n=%d BT=%s S=%s: %.1f us per (matmul + finish), in a captured graph" % (E, n, os.environ.get("HB_MM_FORCE_BT"), os.environ.get("HB_MM_FORCE_S"), e0.elapsed_time(e1) * 1e3 / 20), flush=True)
