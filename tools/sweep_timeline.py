"""Where a train step's streams end, without a profiler: events at the last kernel of the forward, of the full sweep (main stream), of
the aux sweep (its stream) and of the weight-gradient side stream, as ms since the step's first kernel, next to the host's clock at
the moment each was enqueued.   python tools/sweep_timeline.py [--batch 4] [--size 512]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdvit_amd
from mdvit_amd import ops, train
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=args.size, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                        num_domains=4, decoder_name="MLPFM").to(dev).train()
ops.enable_side_stream(True)
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(args.batch, args.size, rank=0, step=s, device=dev) for s in range(2)]


def step(i, evs=None):
    return train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4, phase_events=evs)


for i in range(4):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    # two steps back to back: the second one is the steady state (the host starts it while the GPU still runs the first)
    step(0)
    train._timeline = []
    evs = []
    h0 = time.perf_counter()
    step(1, evs)
    h1 = time.perf_counter()
    tl, train._timeline = train._timeline, None
    torch.cuda.synchronize()
    e0 = evs[0][1]
    print(f"rep {rep}: host enqueue of the step {1e3 * (h1 - h0):.1f} ms")
    rows = [(e0.elapsed_time(e), tag, None) for tag, e in evs] + [(e0.elapsed_time(e), tag, 1e3 * (h - h0)) for tag, e, h in tl]
    for t, tag, h in sorted(rows):
        print(f"   {t:7.2f} ms  {tag:45s}" + (f"  enqueued at host {h:6.1f} ms" if h is not None else ""))
