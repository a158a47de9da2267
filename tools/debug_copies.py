"""Which host-side call sites make device-to-device copies / fills / adds in a train step?  Counts Tensor.clone / copy_ /
contiguous (on non-contiguous) / zero_ / torch.zeros / torch.cat calls by caller.   PYTHONPATH=. python tools/debug_copies.py"""
import collections, traceback
import torch
import mdvit_amd
from mdvit_amd import ops, synthetic
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.train import mdvit_train_step

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                    decoder_name="MLPFM").to(dev).train()
accum = GradAccumulator(m.parameters())
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
ops.enable_side_stream(True)
batches = synthetic.make_step_batches(4, 512, device=dev)
for _ in range(2):
    mdvit_train_step(m, batches, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "mdvit_amd" in fr.filename or "bench" in fr.filename:
            return f"{fr.filename.split('/')[-1]}:{fr.lineno} {fr.name}"
    return "torch-internal (autograd engine)"


def wrap(obj, name, cond=lambda *a, **k: True):
    orig = getattr(obj, name)

    def f(*a, **k):
        if cond(*a, **k):
            counts[f"{name:12s} {site()}"] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


wrap(torch.Tensor, "clone")
wrap(torch.Tensor, "copy_")
wrap(torch.Tensor, "zero_")
wrap(torch.Tensor, "contiguous", lambda t, *a, **k: not t.is_contiguous())
wrap(torch, "zeros")
wrap(torch, "cat")
wrap(torch, "stack")
mdvit_train_step(m, batches, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
torch.cuda.synchronize()
for k, v in counts.most_common(40):
    print(f"{v:4d}  {k}")

# ---- and what the autograd engine itself copies: aten::copy_ / clone / add_ events with their Python stacks
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    mdvit_train_step(m, batches, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::add_", "aten::add", "aten::contiguous", "aten::fill_", "aten::zero_"):
        st = [s for s in ev.stack if "mdvit_amd" in s or "bench" in s]
        parent = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        agg[(ev.name, parent, st[0].split("/")[-1] if st else "(engine)", str(ev.input_shapes)[:60])] += 1
print("aten-level copies / adds / fills by (op, parent op, first repo frame, shapes):")
for k, v in agg.most_common(40):
    print(f"{v:4d}  {k}")
