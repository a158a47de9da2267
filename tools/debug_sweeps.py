"""debug: all-at-once step, aux sweep and uni sweep checked separately against the oracle, repeated"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import mdvit_amd
from mdvit_amd.losses import domain_losses
from oracle import mdvit_ref as R
from oracle.gen_golden import synth_image, synth_label
from oracle.params import make_params

dev = torch.device("cuda:0")
S, B, doms = 64, 2, (0, 1, 2, 3)
pn = make_params(5, model="MDViT", adapt_method="Sup")
cpu_batches = [(synth_image(900 + d, B, S, S), synth_label(910 + d, B, S, S), d) for d in doms]

# oracle: aux-only and uni-only gradients
P = R.to_torch(pn)
leaves = {k: v for k, v in P.items() if v.is_floating_point() and "running_" not in k}
for v in leaves.values():
    v.requires_grad_(True)
tot = tot_aux = tot_kt = 0.0
for img, lab, sid in cpu_batches:
    dl = F.one_hot(torch.full((B,), sid, dtype=torch.long), 4).float()
    out, aux = R.mdvit_forward(P, img, dl, str(sid), R.RefState(training=True))
    l, la, lk = R.domain_losses(out, aux, lab)
    tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
tot_aux.backward(retain_graph=True)
g_aux = {k: (None if v.grad is None else v.grad.clone()) for k, v in leaves.items()}
for v in leaves.values():
    v.grad = None
(0.5 * tot_kt + 0.5 * tot).backward()
g_uni = {k: (None if v.grad is None else v.grad.clone()) for k, v in leaves.items()}

def build():
    m = mdvit_amd.MDViT(img_size=S, adapt_method="Sup")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in pn.items()}, strict=False)
    for i in range(1, 5):
        getattr(m, f"debranch{i}").dropout.p = 0.0
    return m.to(dev).train()

def rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))

def report(tag, m, ref):
    errs = {}
    for n, p in m.named_parameters():
        r = ref[n]
        if r is None or float(r.abs().max()) < 1e-7 or p.grad is None:
            continue
        errs[n] = rel(p.grad.detach().cpu(), r)
    bad = sorted([(n, e) for n, e in errs.items() if not e < 1e-3], key=lambda kv: -kv[1])
    print(tag, "n_bad=", len(bad), [(n[-40:], f"{e:.1e}") for n, e in bad[:4]], flush=True)

batches = [(i.to(dev), l.to(dev), torch.full((B,), d, dtype=torch.long, device=dev)) for i, l, d in cpu_batches]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m = build()
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    tot = tot_aux = tot_kt = 0.0
    for img, lab, sid in batches:
        dl = F.one_hot(sid, 4).float()
        out, aux = m(img, dl, str(int(sid[0])))
        l, la, lk = domain_losses(out, aux, lab)
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    for p in da:
        p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    for p in da:
        p.requires_grad = True
    report(f"{it} aux-sweep", m, g_aux)
    m.zero_grad(set_to_none=True)
    (0.5 * tot_kt + 0.5 * tot).backward()
    report(f"{it} uni-sweep", m, g_uni)
    del m
