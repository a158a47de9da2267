"""debug: gradients of a 2-domain step -- per-domain backward vs all-at-once vs CPU oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import mdvit_amd
from mdvit_amd.train import mdvit_train_step
from oracle import mdvit_ref as R
from oracle.gen_golden import synth_image, synth_label
from oracle.params import make_params

dev = torch.device("cuda:0")
S, B, doms = 64, 2, (0, 1, 2, 3)
pn = make_params(5, model="MDViT", adapt_method="Sup")
cpu_batches = [(synth_image(900 + d, B, S, S), synth_label(910 + d, B, S, S), d) for d in doms]
losses, gref = R.mdvit_train_step(R.to_torch(pn), cpu_batches, R.RefState(training=True))

def build():
    m = mdvit_amd.MDViT(img_size=S, adapt_method="Sup")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in pn.items()}, strict=False)
    for i in range(1, 5):
        getattr(m, f"debranch{i}").dropout.p = 0.0
    return m.to(dev).train()

batches = [(i.to(dev), l.to(dev), torch.full((B,), d, dtype=torch.long, device=dev)) for i, l, d in cpu_batches]
res = {}
def rel(a, b):
    if a is None or b is None:
        return float("nan") if (a is None) != (b is None) else 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))

for it in range(8):
    mode = "per_domain" if it % 2 == 0 else "all_at_once"
    m = build()
    mdvit_train_step(m, batches, per_domain_backward=(mode == "per_domain"))
    torch.cuda.synchronize()
    g = {n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()}
    errs = {n: rel(g[n], gref[n]) for n in g if gref[n] is not None and float(gref[n].abs().max()) > 1e-7}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    nbad = sum(1 for v in errs.values() if not v < 1e-3)
    print(it, mode, "n_bad(>1e-3)=", nbad, "worst:", [(n[-45:], f"{v:.1e}") for n, v in worst], flush=True)
    del m, g
sys.exit(0)

def rel(a, b):
    if a is None or b is None:
        return float("nan") if (a is None) != (b is None) else 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))

bad = 0
for n in res["per_domain"]:
    r = gref[n]
    ea, eb, ec = rel(res["per_domain"][n], r), rel(res["all_at_once"][n], r), rel(res["all_at_once_again"][n], res["all_at_once"][n])
    big = r is not None and float(r.abs().max()) > 1e-7
    if big and (not (ea < 2e-3) or not (eb < 2e-3) or ec > 1e-4):
        bad += 1
        print(f"{n:70s} per_domain_vs_oracle={ea:.2e} all_at_once_vs_oracle={eb:.2e} rerun_diff={ec:.2e} |ref|={float(r.abs().max()):.2e}")
print("mismatching tensors:", bad)
