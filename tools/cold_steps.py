"""Host enqueue time of the first steps of a fresh process (allocator growth, lazy kernel loading, cold caches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.optim import FusedAdamW
from mdvit_amd.parallel import GradAccumulator
from mdvit_amd.synthetic import make_step_batches
from mdvit_amd.train import mdvit_train_step
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NSTEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                        num_domains=4, decoder_name="MLPFM").to(dev).train()
ops.enable_side_stream(True)
accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
accum.attach_sinks()
opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
pool = [make_step_batches(BATCH, 512, rank=0, step=s, device=dev) for s in range(2)]
torch.cuda.synchronize()
for i in range(NSTEPS):
    st = torch.cuda.memory_stats()
    a0 = st.get("num_device_alloc", 0)
    t0 = time.perf_counter()
    mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"step {i:2d}: enqueue {1e3 * (t1 - t0):7.1f} ms  to idle {1e3 * (t2 - t0):7.1f} ms  device allocs +{st.get('num_device_alloc', 0) - a0}  reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB  allocated peak {st['allocated_bytes.all.peak'] / 2**30:.1f} GiB  retries {st.get('num_alloc_retries', 0)}  device frees {st.get('num_device_free', 0)}", flush=True)
