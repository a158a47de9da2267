"""debug: snapshot every tensor autograd saves during domain 0's forward; after the other forwards and
after each sweep, report which saved tensors changed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import mdvit_amd
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

def check(tag, snaps, params):
    nbad = 0
    for i, (t, c, where) in enumerate(snaps):
        if any(t.data_ptr() == p.data_ptr() for p in params):
            continue
        same = torch.equal(t, c) or bool(((t == c) | (t.isnan() & c.isnan())).all())
        if not same:
            diff = (t != c) & ~(t.isnan() & c.isnan())
            idx = diff.reshape(-1).nonzero().reshape(-1)
            nbad += 1
            print(f"  [{tag}] saved tensor #{i} {where} shape={tuple(t.shape)} stride={t.stride()} ptr={t.data_ptr():x} changed elems={int(diff.sum())} first={int(idx[0])} last={int(idx[-1])} old={c.reshape(-1)[idx[0]].item():.4g} new={t.reshape(-1)[idx[0]].item():.4g}", flush=True)
    return nbad

import traceback
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m = build()
    params = list(m.parameters()) + list(m.buffers())
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    snaps = []
    def pack(t):
        if t.is_cuda and t.is_floating_point():
            fr = [f for f in traceback.extract_stack(limit=12) if "mdvit_amd" in f.filename]
            where = "/".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:])
            snaps.append((t, t.clone(), where))
        return t
    tot = tot_aux = tot_kt = 0.0
    for k, (img, lab, sid) in enumerate(batches):
        dl = F.one_hot(sid, 4).float()
        if k == 0:
            with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
                out, aux = m(img, dl, str(int(sid[0])))
                l, la, lk = domain_losses(out, aux, lab)
        else:
            out, aux = m(img, dl, str(int(sid[0])))
            l, la, lk = domain_losses(out, aux, lab)
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    for p in da:
        p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    for p in da:
        p.requires_grad = True
    torch.cuda.synchronize()
    gnow = m.debranch1.linear_fuse[0].weight.grad.detach().clone()
    if it == 0:
        gref0 = gnow
    err = float((gnow - gref0).abs().max() / gref0.abs().max())
    nb = check(f"it{it} after aux sweep", snaps, params)
    print("iteration", it, "snapshots", len(snaps), "changed saved tensors", nb, "grad err vs it0 %.2e" % err, flush=True)
    del m, snaps
