"""python tools/debug_poison.py [fuse] [decoder_name] -- with every op-allocated buffer NaN-filled (MDVIT_POISON=1), report the
autograd nodes whose backward turns clean incoming gradients into NaN ones (= a kernel read a buffer nobody wrote)."""
import os, sys
os.environ["MDVIT_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.losses import domain_losses
dev = torch.device("cuda:0")
torch.manual_seed(0)
fuse = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S, B = 64, 2
m = mdvit_amd.MDViT(img_size=S, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                    decoder_name=sys.argv[2] if len(sys.argv) > 2 else "MLPFM").to(dev).train()
G = fuse
img = torch.randn(G * B, 3, S, S, device=dev)
lab = (torch.rand(G * B, 1, S, S, device=dev) > 0.5).float()
sid = torch.arange(G).repeat_interleave(B)
dl = F.one_hot(sid, 4).float().to(dev)
out, aux = m(img, dl, [str(g) for g in range(G)] if G > 1 else "0")
l, la, lk = domain_losses(out, aux, lab)
loss = l + la + lk
print("forward NaN:", bool(torch.isnan(out).any()), bool(torch.isnan(aux).any()), float(loss))
seen, reported = set(), []
def walk(fn):
    if fn is None or fn in seen:
        return
    seen.add(fn)
    def hook(grad_inputs, grad_outputs, fn=fn):
        bad_in = any(g is not None and torch.isnan(g).any() for g in grad_outputs)       # gradients arriving at the node
        bad_out = any(g is not None and torch.isnan(g).any() for g in grad_inputs)       # gradients it produced
        if bad_out and not bad_in and len(reported) < 10:
            reported.append(fn.name())
            shapes = [None if g is None else tuple(g.shape) for g in grad_inputs]
            print("FIRST NaN produced by", fn.name(), "grad shapes", shapes, "NaN flags", [None if g is None else bool(torch.isnan(g).any()) for g in grad_inputs])
    fn.register_hook(hook)
    for nf, _ in fn.next_functions:
        walk(nf)
sys.setrecursionlimit(100000)
walk(loss.grad_fn)
loss.backward()
bad = [n for n, p in m.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()]
print(len(bad), "gradients with NaN")
