"""python tools/debug_poison_step.py -- the whole bench-style step (domain-batched forward, merged sweeps, side-stream weight
gradients into the buckets, one-launch AdamW) with every op-allocated buffer NaN-filled: parameters must stay finite."""
import os, sys
os.environ["MDVIT_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.train import mdvit_train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
S, B = 64, 2
m = mdvit_amd.MDViT(img_size=S, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                    decoder_name=sys.argv[1] if len(sys.argv) > 1 else "MLPFM").to(dev).train()
accum = GradAccumulator(m.parameters())
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
ops.enable_side_stream(True)
batches = [(torch.randn(B, 3, S, S, device=dev), (torch.rand(B, 1, S, S, device=dev) > 0.5).float(), torch.full((B,), d, dtype=torch.long)) for d in range(4)]
for step in range(3):
    res = mdvit_train_step(m, batches, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4, with_metrics=(step == 2))
torch.cuda.synchronize()
bad = [n for n, p in m.named_parameters() if not torch.isfinite(p).all()]
badb = [n for n, b in m.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
print("losses", {k: float(v) for k, v in res.items() if k.endswith("loss")})
print(len(bad), "non-finite parameters", bad[:8], "|", len(badb), "non-finite buffers", badb[:8])
