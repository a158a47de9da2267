"""debug: capture the gradient arriving at every op output inside domain 0's forward (aux sweep) and
report the first one that differs from a known-good iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.losses import domain_losses
from oracle.gen_golden import synth_image, synth_label
from oracle.params import make_params

dev = torch.device("cuda:0")
S, B, doms = 64, 2, (0, 1, 2, 3)
pn = make_params(5, model="MDViT", adapt_method="Sup")
batches = [(synth_image(900 + d, B, S, S).to(dev), synth_label(910 + d, B, S, S).to(dev), torch.full((B,), d, dtype=torch.long, device=dev)) for d in doms]

def build():
    m = mdvit_amd.MDViT(img_size=S, adapt_method="Sup")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in pn.items()}, strict=False)
    for i in range(1, 5):
        getattr(m, f"debranch{i}").dropout.p = 0.0
    return m.to(dev).train()

CAP = {"on": False, "grads": [], "names": []}
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        out = orig(*a, **k)
        if CAP["on"] and isinstance(out, torch.Tensor) and out.requires_grad:
            idx = len(CAP["names"])
            CAP["names"].append(f"{idx}:{name}{tuple(out.shape)}")
            CAP["grads"].append(None)
            def hook(g, idx=idx):
                if g is not None and CAP["grads"][idx] is None:      # first sweep only
                    CAP["grads"][idx] = g.detach().clone()
            out.register_hook(hook)
        return out
    setattr(ops, name, f)
for n in ("linear", "matmul", "rowdot", "upsample_bilinear", "bn_act", "layer_norm", "mlp_residual", "factor_att", "dwconv3x3", "gconv2_3x3", "conv3x3_dense", "stem_conv"):
    wrap(n)

good = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m = build()
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    CAP.update(on=False, grads=[], names=[])
    tot = tot_aux = tot_kt = 0.0
    for k, (img, lab, sid) in enumerate(batches):
        CAP["on"] = (k == 0)
        out, aux = m(img, F.one_hot(sid, 4).float(), str(int(sid[0])))
        CAP["on"] = False
        l, la, lk = domain_losses(out, aux, lab)
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    for p in da:
        p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    torch.cuda.synchronize()
    grads = [None if g is None else g.cpu() for g in CAP["grads"]]
    pg = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith("debranch1")}
    if good is None:
        good = (grads, pg, list(CAP["names"]))
        print("captured", len(grads), "op outputs", flush=True)
    else:
        bad = []
        for i, (a, b) in enumerate(zip(grads, good[0])):
            if (a is None) != (b is None):
                bad.append((good[2][i], "None-mismatch"))
            elif a is not None:
                e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))
                if e > 1e-4:
                    bad.append((good[2][i], f"{e:.1e}"))
        pb = [(n, f"{float((pg[n] - good[1][n]).abs().max() / max(float(good[1][n].abs().max()), 1e-12)):.1e}") for n in pg
              if float((pg[n] - good[1][n]).abs().max() / max(float(good[1][n].abs().max()), 1e-12)) > 1e-4]
        if len(bad) > 10:
            a, b = grads[169], good[0][169]
            d = (a - b).abs()
            thr = 1e-4 * float(b.abs().max())
            ch = (d.amax(dim=(0, 1, 2)) > thr).nonzero().reshape(-1)
            tok = (d.reshape(-1, d.shape[-1]).amax(dim=1) > thr).nonzero().reshape(-1)
            print("it", it, "BAD: dy(BN in) differs in", ch.numel(), "channels", ch[:8].tolist(), "...", ch[-4:].tolist(), "| tokens", tok.numel(), tok[:6].tolist(), "...", tok[-3:].tolist(),
                  "| ratio sample", (a.reshape(-1, 512)[tok[0], ch[:4]] / b.reshape(-1, 512)[tok[0], ch[:4]]).tolist(), flush=True)
            n170 = float((grads[170] - good[0][170]).abs().max())
            print("    grad at BN out maxdiff", n170, " bn params:", [(n, v) for n, v in pb if "linear_fuse.1" in n], flush=True)
    del m
