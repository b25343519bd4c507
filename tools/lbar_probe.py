"""Lbar = -tril(Kbar A^T) (512x512, contraction 8192) under forced tile / split settings (HB_MM_FORCE_BT / HB_MM_FORCE_S)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for bt, s in ((64, 17), (64, 8), (64, 12), (128, 25), (128, 12), (128, 16), (128, 50)):
        env = dict(os.environ, HB_MM_FORCE_BT=str(bt), HB_MM_FORCE_S=str(s))
        subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from henbun_amd import _lib, hip_ops as H
from ctypes import c_void_p
M, n = 512, 8192
lib = _lib.lib()
ws = torch.empty(64 * M * M, dtype=torch.float32, device="cuda")
C = torch.empty(M, M, dtype=torch.float32, device="cuda")
A = torch.randn(M, n, dtype=torch.float32, device="cuda")
B = torch.randn(M, n, dtype=torch.float32, device="cuda")
def run():
    lib.call("hb_matmul_f32", c_void_p(A.data_ptr()), c_void_p(B.data_ptr()), c_void_p(C.data_ptr()), 1, M, M, n, n, n, M,
             0, 0, 0, 0, 1, -1.0, 0.0, None, 0, 0, H.MM_TRIL_OUT, c_void_p(ws.data_ptr()), ws.numel(), H.stream())
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        run()
    g = H.CapturedGraph()
    g.begin()
    for _ in range(20):
        run()
    g.end()
    g.launch(); st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); g.launch(); g.launch(); e1.record(st); st.synchronize()
print("BT=%s S=%s: %.1f us per (matmul + finish), in a captured graph" % (os.environ.get("HB_MM_FORCE_BT"), os.environ.get("HB_MM_FORCE_S"), e0.elapsed_time(e1) * 1e3 / 40), flush=True)
