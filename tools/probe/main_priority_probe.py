"""Does a HIGH-priority main stream (the critical path of the step) help?  The whole step runs under a high-priority stream; the weight-gradient and
aux-sweep streams stay at normal priority.   python tools/probe/main_priority_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
pool = [make_step_batches(4, 512, rank=0, step=s, device=dev) for s in range(2)]
torch.cuda.synchronize()


def run(n):
    for i in range(n):
        train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)


def timed(n=15):
    run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


hp = torch.cuda.Stream(priority=-1)
print("aux priority", ops._aux_priority, flush=True)
for rnd in range(2):
    print(f"round {rnd}: default-stream main {timed():.2f} ms/step", flush=True)
    hp.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hp):
        t = timed()
    torch.cuda.current_stream().wait_stream(hp)
    print(f"round {rnd}: HIGH-priority main  {t:.2f} ms/step", flush=True)
