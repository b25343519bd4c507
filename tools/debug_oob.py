"""Find the plan step that clobbers a Cholesky info word: eager step-by-step replay with a check after each."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import henbun_amd as hb
from models import ExpertsGPR, svgp_data
tf = hb.tf

E, M, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = 4 * n
np.random.seed(0)
X, Y, Z = svgp_data(N, M, 0, domain=float(M))
ells = list(np.linspace(0.6, 1.2, E)) + list(np.linspace(0.8, 1.4, E))
m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=ells, dtype="float32")
opt = m.ELBO(); opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
print("E0", opt.run(minibatch_size=n), flush=True)
plan = opt._get_plan("opt", n)
sess = m._session
with plan._on_stream():
    st = torch.cuda.current_stream()
    for it in range(3):
        for i, s in enumerate(plan.steps):
            s()
            st.synchronize()
            for info, label in list(plan._infos) + [(i_, "run:" + l_) for pl in opt._plans.values() if pl is not plan for i_, l_ in pl._infos]:
                if info.abs().max().item() != 0:
                    print("iter", it, "step", i, plan.step_labels.get(id(s), "other"), label, info.tolist(), flush=True)
                    sys.exit(0)
        print("iter", it, "clean; theta finite:", bool(torch.isfinite(sess.theta).all()),
              "g finite:", bool(torch.isfinite(plan.gflat).all()), float(plan.gflat.abs().max()), flush=True)

def all_infos():
    return [(l_, i_.tolist()) for pl in opt._plans.values() for i_, l_ in pl._infos]

print("graph mode", flush=True)
def neighbours():
    ptrs = []
    for pl_key, pl in opt._plans.items():
        for t, b in pl._buf.items():
            ptrs.append((b.data_ptr(), b.numel() * b.element_size(), pl_key[0], repr(t)))
        for i_, l_ in pl._infos:
            ptrs.append((i_.data_ptr(), i_.numel() * 4, pl_key[0], "INFO " + l_))
    ptrs.sort()
    for k, p_ in enumerate(ptrs):
        if p_[3].startswith("INFO"):
            for q in ptrs[max(0, k - 4):k + 3]:
                print("   %x +%d %s %s" % q)
            print()

for it in range(60):
    try:
        opt.optimize(maxiter=1, minibatch_size=n)
    except Exception as ex:
        print("EXC", ex)
        print(all_infos())
        neighbours()
        break
    torch.cuda.synchronize()
    bad = [x for x in all_infos() if any(x[1])]
    fin = bool(torch.isfinite(sess.theta).all())
    if bad or not fin or it % 10 == 0:
        print("it", it, "theta finite", fin, "gmax", float(plan.gflat.abs().max()), "infos", bad, flush=True)
    if bad or not fin:
        break
print("E1", opt.run(minibatch_size=n))
