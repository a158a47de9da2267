"""Many consecutive bench steps without any host throttle: loss, allocator state, what the side stream holds, ms per step.   python tools/probe/long_run_probe.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mdvit_amd
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
pool = [make_step_batches(4, 512, rank=0, step=s, device=dev) for s in range(4)]
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
for i in range(N):
    r = train.mdvit_train_step(model, pool[i % 4], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    if i % 60 == 59:
        torch.cuda.synchronize()
        print(i + 1, "steps: loss %.4f  allocated %.2f GiB (max %.2f) reserved %.2f GiB  keepalive %d  side groups %d holding %.2f GiB  %.1f ms/step" % (
              float(r["loss"]), torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
              len(ops._side_keepalive), len(ops._side_groups), ops._side_held[0] / 2**30, 1e3 * (time.perf_counter() - t0) / (i + 1)), flush=True)
