"""Which torch-native ops does one MDViT train step still launch, and from where?  torch.profiler over two steps of the bench
workload; prints the aten ops by count / host time and, for the copy / fill / cat / add family, the Python call sites.
    python tools/host_profile.py [--batch 4]
"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    import mdvit_amd
    from mdvit_amd import ops
    from mdvit_amd.optim import FusedAdamW
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.synthetic import make_step_batches
    from mdvit_amd.train import mdvit_train_step
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = mdvit_amd.MDViT(img_size=args.size, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                            num_domains=4, decoder_name="MLPFM").to(dev).train()
    ops.enable_side_stream(True)
    accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
    accum.attach_sinks()
    opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
    pool = [make_step_batches(args.batch, args.size, rank=0, step=s, device=dev) for s in range(2)]
    for i in range(3):
        mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    nsteps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(nsteps):
            mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
        torch.cuda.synchronize()
    evs = prof.events()
    by = collections.defaultdict(lambda: [0, 0.0])
    sites = collections.defaultdict(lambda: collections.Counter())
    watch = ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat", "aten::add", "aten::add_", "aten::stack",
             "aten::contiguous", "aten::to", "aten::_to_copy", "aten::mul", "aten::div", "aten::neg", "aten::sum", "aten::empty", "aten::one_hot",
             "aten::zeros_like", "aten::rand", "aten::lt", "aten::_foreach_add_", "aten::select", "aten::view", "aten::reshape", "aten::slice", "aten::narrow", "aten::split")
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
            continue
        by[e.name][0] += 1; by[e.name][1] += e.self_cpu_time_total
        if e.name in watch and e.stack:
            fr = [f for f in e.stack if "/mdvit_amd/" in f or "bench" in f or "autograd" in f]
            sites[e.name][(fr[0] if fr else e.stack[0])[-110:]] += 1
    print(f"aten ops per step (host self time, us per step), {nsteps} steps profiled")
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {n / nsteps:8.1f} calls  {t / nsteps:9.1f} us  {k}")
    for k in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::add_", "aten::zeros", "aten::clone", "aten::_to_copy", "aten::mul", "aten::stack", "aten::contiguous"):
        if k in sites:
            print(f"{k}: call sites (count per step)")
            for s, n in sites[k].most_common(14):
                print(f"    {n / nsteps:6.1f}  {s}")
    # device-side: memcpy kinds
    mem = collections.Counter()
    for e in evs:
        if e.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in e.name or "Memset" in e.name or "copyBuffer" in e.name):
            mem[e.name[:60]] += 1
    print("device copies per step:", {k: v / nsteps for k, v in mem.items()})


main()
