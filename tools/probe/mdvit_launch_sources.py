"""Which host call sites issue the small launches of a MDViT bs=4 train step (hipMemcpyAsync = __amd_rocclr_copyBuffer, fills, adds)?
torch.profiler with Python stacks over one step of the default `bench.py` step.   python tools/probe/mdvit_launch_sources.py"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
import mdvit_amd
from mdvit_amd.train import mdvit_train_step
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM").to(dev).train()
ops.enable_side_stream(True)
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(4, 512, rank=0, step=s, device=dev) for s in range(2)]
for i in range(3):
    mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    mdvit_train_step(model, pool[1], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events())
print("events mentioning copies, sets, fills:")
for n, c in names.most_common():
    if any(s in n.lower() for s in ("memcpy", "memset", "copy", "fill", "zero", "aten::add", "aten::cat", "aten::stack", "aten::mul", "aten::sum")):
        print(f"{c:5d}  {n}")
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::stack", "aten::_to_copy", "aten::clone", "aten::mul", "aten::sum"):
        site = "(engine)"
        for fr in (ev.stack or []):
            if "mdvit_amd" in fr or "bench.py" in fr:
                site = fr.strip()[-80:]
                break
        par = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        agg[(ev.name, par[:50], site)] += 1
for k, v in agg.most_common(45):
    print(f"{v:4d}  {k}")
