"""Host-side cost of one MDViT train step: cProfile over a few steps of the bench workload (kernels run asynchronously, so this is
the enqueue path: Python glue, autograd, allocator, ctypes, hipLaunchKernel).   python tools/host_cprofile.py [--steps 5]"""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--model", choices=["mdvit", "transfuse"], default="mdvit", help="transfuse: TransFuse_S_adapt at 256 x 256 (use --batch 8)")
    args = ap.parse_args()
    import mdvit_amd
    from mdvit_amd import ops
    from mdvit_amd.optim import FusedAdamW
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.synthetic import make_step_batches
    from mdvit_amd.train import mdvit_train_step
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tf = args.model == "transfuse"
    if tf:
        from mdvit_amd.transfuse import TransFuse_S_adapt, transfuse_train_step
        ops.reserve_streams(side=True, sweep=False, branch=True)
        model = TransFuse_S_adapt(num_classes=1, drop_rate=0.2, pretrained=False, num_domains=4).to(dev).train()
    else:
        model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                                num_domains=4, decoder_name="MLPFM").to(dev).train()
    ops.enable_side_stream(True)
    accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
    accum.attach_sinks()
    opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
    pool = [make_step_batches(args.batch, 256 if tf else 512, rank=0, step=s, device=dev) for s in range(2)]

    def run(n):
        for i in range(n):
            if tf:
                transfuse_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, fuse_domains=True)
            else:
                mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4)
    run(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(args.steps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"unprofiled: enqueue {1e3 * (t1 - t0) / args.steps:.1f} ms/step, step {1e3 * (t2 - t0) / args.steps:.1f} ms")
    torch.autograd.set_multithreading_enabled(False)        # backward nodes run on THIS thread: the profile sees them
    t0 = time.perf_counter(); run(args.steps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"single-threaded autograd: enqueue {1e3 * (t1 - t0) / args.steps:.1f} ms/step, step {1e3 * (t2 - t0) / args.steps:.1f} ms")
    pr = cProfile.Profile()
    pr.enable(); run(args.steps); pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(60)
    st.sort_stats("cumulative").print_stats(70)


main()
