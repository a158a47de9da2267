"""Which host-side call sites make device-to-device copies in a train step?  (counts _c() conversions of non-contiguous
tensors by caller, and torch-level copy kernels via the profiler)   PYTHONPATH=. python tools/debug_copies.py"""
import collections, traceback
import torch
import mdvit_amd
from mdvit_amd import ops, synthetic
from mdvit_amd.train import mdvit_train_step

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                    decoder_name="MLPFM").to(dev).train()
batches = synthetic.make_step_batches(4, 512, device=dev)
for _ in range(2):
    mdvit_train_step(m, batches, optimizer=None, merged_sweeps=True, fuse_domains=4)
    m.zero_grad(set_to_none=True)
counts = collections.Counter()
orig = ops._c


def counting_c(t):
    if not t.is_contiguous():
        fr = traceback.extract_stack(limit=3)[0]
        counts[f"{fr.name}:{fr.lineno} shape={tuple(t.shape)} stride={t.stride()}"] += 1
    return orig(t)


ops._c = counting_c
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    mdvit_train_step(m, batches, optimizer=None, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
ops._c = orig
print("non-contiguous -> contiguous conversions in ops._c:")
for k, v in counts.most_common(30):
    print(f"  {v:4d}  {k}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
