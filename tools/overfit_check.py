"""End-to-end sanity: 40 optimisation steps on ONE fixed 4-domain batch must drive the losses down (fused AdamW, merged sweeps,
domain-batched forward, side-stream weight gradients).   PYTHONPATH=. python tools/overfit_check.py [size] [batch]"""
import sys, torch
import mdvit_amd
from mdvit_amd import ops, synthetic
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.train import mdvit_train_step
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = mdvit_amd.MDViT(img_size=S, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                    decoder_name="MLPFM").to(dev).train()
accum = GradAccumulator(m.parameters())
opt = FusedAdamW(accum, lr=3e-4, weight_decay=0.05)
ops.enable_side_stream(True)
batches = synthetic.make_step_batches(B, S, device=dev)
hist = []
for step in range(40):
    r = mdvit_train_step(m, batches, optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4, with_metrics=(step % 13 == 0))
    hist.append((float(r["loss"]), float(r["aux_loss"]), float(r["kt_loss"])))
    if step % 13 == 0 or step == 39:
        print(step, ["%.4f" % v for v in hist[-1]], {k: [round(float(x), 3) for x in v.flatten()[:4]] for k, v in r.items() if k == "metrics"})
assert all(torch.isfinite(p).all() for p in m.parameters())
assert hist[-1][0] < 0.7 * hist[0][0] and hist[-1][1] < 0.7 * hist[0][1], (hist[0], hist[-1])
print("ok: loss %.3f -> %.3f, aux %.3f -> %.3f, peak memory %.2f GB" % (hist[0][0], hist[-1][0], hist[0][1], hist[-1][1], torch.cuda.max_memory_allocated() / 1e9))
