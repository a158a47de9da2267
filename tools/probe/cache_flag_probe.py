import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mdvit_amd
from mdvit_amd import ops, train
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM").to(dev).train()
ops.enable_side_stream(True)
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(4, 512, rank=0, step=s, device=dev) for s in range(2)]
orig = ops.take_cache_fill_flag
log = []
def spy():
    f = orig(); log.append(f); return f
ops.take_cache_fill_flag = spy
train.ops.take_cache_fill_flag = spy
for i in range(5):
    log.clear()
    train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    print("step", i, "flags read:", log)
