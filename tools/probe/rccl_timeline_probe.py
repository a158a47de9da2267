"""tools/sweep_timeline.py under a 1-rank RCCL process group with the collectives forced on (what a rank of the data-parallel job enqueues): does RCCL's own
stream push the aux sweep behind the full sweep (a fifth stream on four hardware queues)?   python tools/probe/rccl_timeline_probe.py"""
import os, sys, time
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29587"), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.cuda.set_device(0)
import mdvit_amd
from mdvit_amd import ops, train
if os.environ.get("RESERVE", "1") != "0":
    ops.reserve_streams()            # BEFORE the communicator: the step's streams bind their hardware queues first
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for force in (False, True):
    ops._force_collectives = force
    model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM").to(dev).train()
    ops.enable_side_stream(True)
    accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
    accum.attach_sinks()
    opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
    pool = [make_step_batches(4, 512, rank=0, step=s, device=dev) for s in range(2)]

    def step(i, evs=None):
        return train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4, phase_events=evs)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    # steady-state step time (events around 10 steps) next to the timeline of one step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    for rep in range(2):
        step(0)
        train._timeline = []
        evs = []
        step(1, evs)
        torch.cuda.synchronize()
        tl, train._timeline = train._timeline, None
    print(f"steady state: {ms:.2f} ms per step")
    e0 = evs[0][1]
    rows = [(e0.elapsed_time(e), tag) for tag, e in evs[1:]] + [(e0.elapsed_time(e), tag) for tag, e, _ in tl]
    print(f"collectives forced: {force}   overlapped buckets: {getattr(accum, 'overlapped_buckets', None)}")
    for t, tag in sorted(rows):
        print(f"   {t:8.2f} ms  {tag}")
    del model, accum, opt
dist.destroy_process_group()
