"""Reserved / allocated memory of the bs=32 step over consecutive steps (allocator growth?).   python tools/probe/bs32_memory_probe.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mdvit_amd
from mdvit_amd import ops, train
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM").to(dev).train()
if os.environ.get('PROBE_SIDE', '1') != '0':
    ops.enable_side_stream(True)
if os.environ.get('PROBE_TWO', '1') == '0':
    train._two_stream_sweeps = False
INFL = int(os.environ.get('PROBE_INFLIGHT', '2'))
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(32, 512, rank=0, step=s, device=dev) for s in range(2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ev = []
t_run = None
for i in range(n):
    if i == 3:
        torch.cuda.synchronize(); t_run = time.perf_counter()
    while len(ev) >= INFL:
        ev.pop(0).synchronize()
    train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    e = torch.cuda.Event(); e.record(); ev.append(e)
    print(i, "allocated %.1f GiB  max allocated %.1f  reserved %.1f GiB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30,
                                                                             torch.cuda.memory_reserved() / 2**30),
          "inactive split %.1f GiB  segments %d  retries %d" % (torch.cuda.memory_stats()["inactive_split_bytes.all.current"] / 2**30,
                                                                   torch.cuda.memory_stats()["segment.all.current"], torch.cuda.memory_stats()["num_alloc_retries"]), flush=True)
torch.cuda.synchronize()
if t_run is not None and n > 3:
    print("steps 3..%d: %.1f ms per step, %.1f images/s  (MDVIT_SIDE_HOLD_GIB=%s, run-ahead %d)" % (n - 1, 1e3 * (time.perf_counter() - t_run) / (n - 3),
          128 * (n - 3) / (time.perf_counter() - t_run), os.environ.get("MDVIT_SIDE_HOLD_GIB", "default"), INFL), flush=True)
import collections
for tag in ('final',):
    snap = torch.cuda.memory_snapshot()
    per = collections.defaultdict(lambda: [0, 0, 0])
    for seg in snap:
        st = seg.get('stream', 0)
        per[st][0] += seg['total_size']; per[st][1] += seg['allocated_size']; per[st][2] += 1
    for st, (tot, alloc, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print('stream', st, 'reserved %.1f GiB in %d segments, allocated now %.1f GiB' % (tot / 2**30, n, alloc / 2**30))
    big = sorted(snap, key=lambda s_: -s_['total_size'])[:12]
    print('largest segments (GiB):', [round(s_['total_size'] / 2**30, 2) for s_ in big])
    sizes = collections.Counter(round(s_['total_size'] / 2**30, 2) for s_ in snap if s_['total_size'] > 2**30)
    print('segment sizes > 1 GiB (size: count):', dict(sorted(sizes.items(), key=lambda kv: -kv[0])))
