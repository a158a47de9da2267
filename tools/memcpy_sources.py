"""Which host call sites issue hipMemcpyAsync / hipMemsetAsync in a train step (the `__amd_rocclr_copyBuffer` / fill launches of the
kernel trace)?  torch.profiler (CPU + device activities, Python stacks) over one step of the bench workload.
python tools/memcpy_sources.py [transfuse]"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
from mdvit_amd.train import mdvit_train_step
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
tf = len(sys.argv) > 1 and sys.argv[1] == "transfuse"
if tf:
    from mdvit_amd.transfuse import TransFuse_S_adapt, transfuse_train_step
    ops.reserve_streams(side=True, sweep=False, branch=True)
    model = TransFuse_S_adapt(num_classes=1, drop_rate=0.2, pretrained=False, num_domains=4).to(dev).train()
else:
    model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                            num_domains=4, decoder_name="MLPFM").to(dev).train()
ops.enable_side_stream(True)
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(8 if tf else 4, 256 if tf else 512, rank=0, step=s, device=dev) for s in range(2)]


def step(b):
    if tf:
        transfuse_train_step(model, b, optimizer=opt, accumulator=accum, fuse_domains=True)
    else:
        mdvit_train_step(model, b, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)


for i in range(3):
    step(pool[i % 2])
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(pool[1])
    torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events())
print("runtime / op events mentioning copies, sets, fills:")
for n, c in names.most_common():
    if any(s in n.lower() for s in ("memcpy", "memset", "copy", "fill", "zero", "aten::add", "aten::cat", "aten::stack")):
        print(f"{c:5d}  {n}")
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::stack", "aten::_to_copy", "aten::clone"):
        site = "(engine)"
        for fr in (ev.stack or []):
            if "mdvit_amd" in fr or "bench.py" in fr or "memcpy_sources" in fr:
                site = fr.strip()[-70:]
                break
        par = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        agg[(ev.name, par[:50], site)] += 1
for k, v in agg.most_common(40):
    print(f"{v:4d}  {k}")
