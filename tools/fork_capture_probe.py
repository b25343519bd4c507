"""Two-branch stream capture with the library's own launches (round-2 review item 8: "find the cause of the
hipStreamEndCapture segfault recorded in round 1 before building on fork-capture").

The round-1 experiment list-scheduled independent plan steps onto side streams inside the capture and died inside
hipStreamEndCapture on the batched-experts plan.  This probe captures, through the same C ABI (hb_graph_begin_capture /
hb_graph_end_capture, thread-local capture mode), a graph with a forked branch:

    s1: A = x * 2 ----------------.--> C = A + B (+ a compiled elementwise program, + a GEMM with side jobs pending)
          \\ event e1               / event e2
    s2:     B = matmul(y, y) -----'

  joined   : s1 waits for e2 before the capture ends          -> must replay correctly
  unjoined : the capture ends while s2's work was never joined -> run in a CHILD process: HIP documents
             hipErrorStreamCaptureUnjoined for this; a crash here is the round-1 symptom

`python tools/fork_capture_probe.py` prints one line per case.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(case):
    import torch
    from henbun_amd import _lib
    from henbun_amd import hip_ops as H

    L = _lib.lib()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.randn(1 << 16, device="cuda")
    y = torch.randn(512, 512, device="cuda")
    A, B, C = torch.empty_like(x), torch.empty_like(y), torch.empty_like(x)
    # warm-up outside the capture (workspaces, compiled programs)
    for s in (s1, s2):
        with torch.cuda.stream(s):
            H.ewise("AFFINE", [x], params=[2.0, 0.0], out=A)
            H.matmul(y, y, out=B)
            H.ewise("ADD", [A, B.reshape(-1)[: x.numel()]], out=C)
    torch.cuda.synchronize()
    import ctypes
    exec_ = ctypes.c_void_p(None)
    with torch.cuda.stream(s1):
        L.call("hb_graph_begin_capture", s1.cuda_stream)
        H.ewise("AFFINE", [x], params=[2.0, 0.0], out=A)
        e1 = torch.cuda.Event()
        e1.record(s1)
        s2.wait_event(e1)                      # fork: s2 joins the capture
        with torch.cuda.stream(s2):
            H.matmul(y, y, out=B)              # (matmul_wgk: the launch that hosts side jobs)
            e2 = torch.cuda.Event()
            e2.record(s2)
        if case == "joined":
            s1.wait_event(e2)                  # join
        H.ewise("ADD", [A, B.reshape(-1)[: x.numel()]], out=C)
        try:
            L.call("hb_graph_end_capture", s1.cuda_stream, ctypes.byref(exec_))
        except _lib.HipBackendError as err:
            print("%s: hb_graph_end_capture returned an error (no crash): %s" % (case, err))
            return 0
    for _ in range(3):
        C.zero_()
        with torch.cuda.stream(s1):
            L.call("hb_graph_launch", exec_, s1.cuda_stream)
        torch.cuda.synchronize()
    ref = 2.0 * x + (y @ y).reshape(-1)[: x.numel()]
    err = float((C - ref).abs().max() / ref.abs().max())
    print("%s: captured and replayed 3 times, relative error of C against torch %.2e" % (case, err))
    return 0 if err < 1e-4 else 1


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(run(sys.argv[1]))
    rc = 0
    for case in ("joined", "unjoined"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), case], capture_output=True, text=True, timeout=300)
        out = [l for l in (p.stdout + p.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
        tail = out[-1] if out else ""
        if p.returncode < 0 or p.returncode > 1:
            print("%s: CHILD DIED (return code %d) -- %s" % (case, p.returncode, tail))
        else:
            print(tail)
        rc |= (case == "joined" and p.returncode != 0)
    sys.exit(rc)
