#!/usr/bin/env python3
"""Headline benchmark: MDViT (adapt_method='Sup', MLPFM peer heads) train step at 512x512 on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--size S] [--model mdvit|base]

One step = for each of the 4 domains a forward of B synthetic images, the fused BCE/Dice/KT losses and
the reference's two-sweep backward (multi_train_MDViT.py:129-207), gradient all-reduce across ranks,
AdamW update.  drop_rate = drop_path_rate = 0.1 as in multi_train_MDViT.py:59.  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line (see README/DESIGN for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak; a bf16x3 GEMM issues 3 bf16 MFMAs per product
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=5)      # (the allocator, the derived-weight caches and the GEMM planner settle over the first 3-4 steps)
    ap.add_argument("--batch", type=int, default=4, help="images per domain per GPU (BASELINE configs[1]: 4)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", choices=["mdvit", "mdvit_dsn", "base", "transfuse"], default="mdvit",
                    help="mdvit_dsn: MDViT_DSN, domain-specific norms; transfuse: TransFuse_S_adapt (BASELINE configs[4]: use --batch 8 --size 256)")
    ap.add_argument("--decoder", choices=["MLPFM", "MLP", "Transformer", "DeepLabV3"], default="MLPFM", help="peer heads (MDViT decoder_name); the headline config is MLPFM")
    ap.add_argument("--host-inputs", action="store_true",
                    help="PCIe-inclusive variant: every step's images (uint8 HWC) and labels (uint8) start in pinned host memory and cross to the "
                         "device inside the timed region (the headline number keeps its inputs resident in HBM)")
    ap.add_argument("--no-side-stream", action="store_true")
    ap.add_argument("--precision", choices=["bf16x3", "fp32", "bf16"], default="bf16x3",
                    help="GEMM arithmetic: bf16x3 = fp32 operands split hi+lo into bf16, 3 bf16 MFMAs per product, fp32 accumulate (~1e-5 rel: the "
                         "parity mode and the headline); fp32 = fp32-input MFMA; bf16 = the speed mode: GEMM operands rounded to ONE bf16 plane, one "
                         "MFMA per product, fp32 accumulate, fp32 norms / softmax statistics / losses (drifts ~1e-2: never the parity mode)")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of the one-launch HIP AdamW")
    ap.add_argument("--graph", action="store_true", help="capture the whole step in a HIP graph and replay it")
    ap.add_argument("--fuse-images", type=int, default=128,
                    help="domain batches are fused into one domain-batched forward while the fused batch stays <= this many images (0: one forward per domain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reference-sweeps", action="store_true", help="run the reference's literal two full sweeps instead of the merged (linear-algebra-equivalent) form")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--settle-seconds", type=float, default=60.0,
                    help="untimed settling before the warm-up steps: repeat the step until the host's enqueue time is stable (a fresh box pages the image in); 0: off")
    ap.add_argument("--by-shape", action="store_true", help="key the GEMM event table by (variant, M, N, K) -- for tools/gemm_shapes.py")
    ap.add_argument("--detail", default="", help="write the per-kernel table to this JSON file")
    ap.add_argument("--max-inflight", type=int, default=0,
                    help="steps the host may have enqueued ahead of the GPU (a training loop that reads its loss every step has 1-2).  0 (default): 2 below "
                         "batch 16, 1 from batch 16 up -- there the host needs 22 ms for a 250 ms step, and every step of run-ahead keeps one more step's "
                         "cross-stream tensors in the RESERVED pool (bs=32: 105.9 GiB reserved with 1, 116.5 with 2; peak allocated 86.9 GiB either way since "
                         "ops._side_protect bounds what the weight-gradient stream holds)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the bs=32 leg and the MHSA+DA block roofline at bs=32 that the default N=1 run appends to its JSON line")
    args = ap.parse_args()
    if args.max_inflight <= 0:
        args.max_inflight = 2 if args.batch < 16 else 1
    return args


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """physical cores of this host (sockets x cores per socket), falling back to the logical count"""
    try:
        seen = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(size: int, budget_s: float = 40.0):
    """SURVEY 8(d) protocol on the host cores: the CPU oracle (oracle/mdvit_ref.py, pure torch fp32 -- the same math as
    the reference) for (1) BASELINE configs[0]: BASE bs=4, one domain, and (2) the MDViT sample of the headline workload
    (one domain x one image): SURVEY 8(d)'s 3 warm-up + 5 timed steps, MEDIAN, forward / backward ms split (~55 s per leg on the pool's
    EPYC 9575F).  Both legs are bounded by `budget_s` seconds each: on a host where 8 steps would not fit, the step count falls back
    to 1 warm-up + 3 timed and the JSON says so."""
    import statistics
    import torch
    from oracle import mdvit_ref as R
    from oracle.params import make_params
    from mdvit_amd.synthetic import make_domain_batch
    phys = _physical_cores()
    # SURVEY 8(d) says "all physical cores".  Measured on the pool's 2 x 64-core EPYC 9575F (round 4, profiles/r04c_bench_bs4.json): with 128 threads the
    # oracle's step takes 19.4 s, with 64 threads 7.0 s -- oneDNN's thread pool collapses past one socket.  `value` is therefore the FASTER of the two (the
    # fairer baseline), measured at min(physical, 64) threads; the all-cores figure is reported next to it (`mdvit_1img_all_physical_cores`).
    cores = min(phys, 64)
    torch.set_num_threads(cores)

    def timed(step, imgs):
        t0 = time.perf_counter(); step(); first = time.perf_counter() - t0          # warm-up 1 (cold: allocator, oneDNN primitives)
        n_warm = 3 if first * 8 <= budget_s else 1
        n_timed = 5 if first * 8 <= budget_s else 3
        for _ in range(n_warm - 1):
            step()
        ts, fw, bw = [], [], []
        for _ in range(n_timed):
            t0 = time.perf_counter()
            f_ms, b_ms = step()
            ts.append(time.perf_counter() - t0); fw.append(f_ms); bw.append(b_ms)
        med = statistics.median(ts)
        return {"images_per_s": round(imgs / med, 4), "median_step_s": round(med, 3), "fwd_ms": round(statistics.median(fw), 1),
                "bwd_ms": round(statistics.median(bw), 1), "warmup": n_warm, "timed": n_timed, "first_cold_step_s": round(first, 3)}

    # (2) MDViT Sup, one domain x one image
    P = R.to_torch(make_params(0, model="MDViT", adapt_method="Sup"))
    img, lab, _ = make_domain_batch(1, size, 0, 1234)

    def mdvit_step():
        st = R.RefState(training=True, drop_rate=0.1, drop_path_rate=0.1, aux_drop=0.1)
        tm = {}
        R.mdvit_train_step(P, [(img, lab, 0)], st, timing=tm)
        return tm.get("fwd_ms", 0.0), tm.get("bwd_ms", 0.0)

    mdvit = timed(mdvit_step, 1)
    mdvit_all = None
    if phys > cores:
        torch.set_num_threads(phys)
        t0 = time.perf_counter(); mdvit_step(); cold = time.perf_counter() - t0
        t0 = time.perf_counter(); f_ms, b_ms = mdvit_step(); warm = time.perf_counter() - t0
        mdvit_all = {"threads": phys, "images_per_s": round(1.0 / warm, 4), "step_s": round(warm, 3), "fwd_ms": round(f_ms, 1), "bwd_ms": round(b_ms, 1),
                     "warmup": 1, "timed": 1, "first_cold_step_s": round(cold, 3)}
        torch.set_num_threads(cores)
    # (3) the headline workload's own shape (BASELINE configs[1]): 4 domains x bs=4 -- ONE step (the oracle's primitives are warm from the legs above; a second
    # step would put the whole CPU baseline past two minutes)
    cfg2 = None
    try:
        batches = [make_domain_batch(4, size, d, 1234) for d in range(4)]
        st2 = R.RefState(training=True, drop_rate=0.1, drop_path_rate=0.1, aux_drop=0.1)
        tm2 = {}
        t0 = time.perf_counter()
        R.mdvit_train_step(P, [(b_[0], b_[1], d_) for d_, b_ in enumerate(batches)], st2, timing=tm2)
        dt2 = time.perf_counter() - t0
        cfg2 = {"images_per_s": round(16 / dt2, 4), "step_s": round(dt2, 3), "fwd_ms": round(tm2.get("fwd_ms", 0.0), 1), "bwd_ms": round(tm2.get("bwd_ms", 0.0), 1),
                "warmup": 0, "timed": 1, "threads": cores,
                "sample": f"BASELINE configs[1] shape: MDViT Sup, 4 domains x bs=4 {size}x{size}, fwd + BCE/Dice/KT + two-sweep bwd, one step"}
        del batches
    except Exception as e:           # the baseline is a report, never a reason to lose the bench line
        cfg2 = {"error": repr(e)}
    del P
    # (1) BASELINE configs[0]: BASE (no DA, no MKD) bs=4, single domain
    PB = R.to_torch(make_params(0, model="BASE", adapt_method=False))
    imgb, labb, _ = make_domain_batch(4, size, 0, 1234)

    def base_step():
        st = R.RefState(training=True, drop_rate=0.1, drop_path_rate=0.1, aux_drop=0.1)
        tm = {}
        R.base_train_step(PB, imgb, labb, None, st, timing=tm)
        return tm.get("fwd_ms", 0.0), tm.get("bwd_ms", 0.0)

    base = timed(base_step, 4)
    # `value` = the leg with the GPU headline's own workload (BASELINE configs[1]: 4 domains x bs=4) when it ran; the 1-image leg otherwise (the 1-image
    # figure is the SLOWER of the two per image -- 0.137 against 0.233 images/s on the same 64 threads)
    head = cfg2 if (cfg2 and "images_per_s" in cfg2) else None
    return {"value": head["images_per_s"] if head else mdvit["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(), "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "sample": (head["sample"] + f", fp32 torch CPU oracle, {head['step_s']} s") if head else
                      f"MDViT Sup: 1 domain x 1 image {size}x{size}, fwd + BCE/Dice/KT + two-sweep bwd, fp32 torch CPU oracle, "
                      f"{mdvit['warmup']} warm-up + {mdvit['timed']} timed steps, median {mdvit['median_step_s']} s",
            "mdvit_1img": mdvit, "mdvit_1img_all_physical_cores": mdvit_all, "mdvit_cfg2_4x4": cfg2,
            "base_bs4": dict(base, sample=f"BASELINE configs[0]: BASE bs=4 {size}x{size}, 1 domain, fwd + BCE/Dice + bwd, "
                                          f"{base['warmup']} warm-up + {base['timed']} timed steps, median")}


def _free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _spawn_ranks(n: int) -> int:
    """one child job (N rank processes) via torch.distributed.run; this parent never initialises the GPU"""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _child_json(cmd, timeout):
    """run a helper leg as a CHILD process (never exec: this process holds the GPU) and return its last JSON line"""
    import subprocess
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"no JSON from {' '.join(cmd[1:])} (rc {r.returncode}): {r.stderr[-300:]}"}
    except Exception as e:
        return {"error": repr(e)}


def _rocprof_avg_us(name, model="mdvit"):
    """the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of THIS leg's command (profiles/): the
    event-timed figure also holds the time a launch waits for CUs that the other streams' kernels occupy, rocprofv3's does not.
    (Round 4: the TransFuse leg used to read the MDViT summary -- the same kernel name over other shapes: its frac_rocprof was not about its launches.)"""
    import csv, glob
    pattern = {"mdvit": "r*_bench_bs4_kernel_stats.csv", "transfuse": "r*_transfuse_bs8_kernel_stats.csv"}.get(model)
    if pattern is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    want = name.split("+")[0].replace(" ", "")
    try:
        with open(files[-1]) as f:
            for r in csv.DictReader(f):
                if want in r["Name"].replace("(anonymous namespace)::", "").replace(" ", ""):
                    return round(float(r["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    return None


def _trace_largest(model="mdvit"):
    """the largest kernel by total time in the committed rocprofv3 --kernel-trace --stats summary of this leg's command, over every kernel family"""
    import csv, glob
    pattern = {"mdvit": "r*_bench_bs4_kernel_stats.csv", "transfuse": "r*_transfuse_bs8_kernel_stats.csv"}.get(model)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))) if pattern else []
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rows = list(csv.DictReader(f))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        return {"name": r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:120], "share_of_kernel_time": round(float(r["TotalDurationNs"]) / tot, 4),
                "avg_launch_us": round(float(r["AverageNs"]) / 1e3, 2), "source": "profiles/" + os.path.basename(files[-1])}
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed as is: start one fresh process per GPU (torch.distributed.run, rendezvous on
        # 127.0.0.1) BEFORE anything in this process touches the GPU, relay rank 0's JSON line, exit with the job's code.
        sys.exit(_spawn_ranks(args.gpus))
    if os.environ.get("MDVIT_BENCH_DRYRUN"):
        # launch-path check without a GPU (tests/test_host.py): the ranks this command line produces rendezvous over gloo on
        # 127.0.0.1, count themselves, and rank 0 prints the line the driver would parse
        dist.init_process_group("gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": args.gpus, "world": world, "ranks_seen": int(ones.item()), "steps": args.steps, "warmup": args.warmup}), flush=True)
        dist.barrier(); dist.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or plain `python bench.py --gpus N`)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import mdvit_amd
    from mdvit_amd import ops
    if world > 1:
        # the step's own streams bind their hardware queues BEFORE RCCL's communicator brings its streams (four queues: see ops.reserve_streams)
        if not args.no_side_stream:
            ops.reserve_streams(side=True, sweep=args.model in ("mdvit", "mdvit_dsn"), branch=args.model == "transfuse")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    from mdvit_amd.parallel import GradAccumulator, broadcast_parameters
    from mdvit_amd.synthetic import make_step_batches
    from mdvit_amd.train import base_train_step, mdvit_train_step

    if os.environ.get("MDVIT_AUTOGRAD_MT", "1") == "0":
        torch.autograd.set_multithreading_enabled(False)
    ops.set_gemm_precision(args.precision)
    # useful-flop roof of the GEMM arithmetic in use: fp32 MFMA peak, or a third of the bf16 peak (3 MFMAs per product)
    peak_mfma = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0, "bf16": PEAK_BF16_MFMA_TFLOPS}[args.precision]
    torch.manual_seed(0)
    if args.model in ("mdvit", "mdvit_dsn"):
        cls = mdvit_amd.MDViT if args.model == "mdvit" else mdvit_amd.MDViT_DSN
        model = cls(img_size=args.size, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d,
                    adapt_method="Sup", num_domains=4, decoder_name=args.decoder).to(dev).train()
        domains, flop_per_img = (0, 1, 2, 3), 251.0e9
    elif args.model == "transfuse":
        from mdvit_amd.transfuse import TransFuse_S_adapt, transfuse_train_step
        if args.size != 256:
            raise SystemExit("TransFuse_S_adapt accepts 256x256 inputs only (DeiT.py:134): use --size 256")
        model = TransFuse_S_adapt(num_classes=1, drop_rate=0.2, pretrained=False, num_domains=4).to(dev).train()      # multi_train_TransFuse.py's constructor defaults
        domains, flop_per_img = (0, 1, 2, 3), 71.1e9          # ~11.86 GMAC forward per 256x256 image, x3 for the train step
    else:
        model = mdvit_amd.BASE(drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method=False).to(dev).train()
        domains, flop_per_img = (0,), 137.7e9
    broadcast_parameters(model)
    if not args.no_side_stream:
        ops.enable_side_stream(True)      # wgrad kernels overlap the dgrad chain and add straight into the gradient buckets
    # fused accumulation; with world > 1 every bucket but the domain adapters' is all-reduced underneath the aux sweep
    accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
    accum.attach_sinks()                              # wgrad GEMMs add straight into the gradient buckets
    if args.torch_adamw:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True, capturable=args.graph)
    else:
        from mdvit_amd.optim import FusedAdamW
        opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)      # ONE launch over all 432 parameter tensors
    # a small pool of distinct synthetic steps, resident in HBM before timing
    pool = [make_step_batches(args.batch, args.size, rank=rank, step=s, device=dev, domains=domains) for s in range(2)]

    fuse = max(1, min(len(domains), args.fuse_images // max(1, args.batch))) if args.model in ("mdvit", "mdvit_dsn") else 1

    drift = None
    if args.precision == "bf16" and args.model == "mdvit":
        # what the speed mode costs in accuracy: the same weights and images through the parity arithmetic and through bf16
        import torch.nn.functional as F
        img0 = pool[0][0][0][:2]
        dl0 = F.one_hot(torch.zeros(img0.shape[0], dtype=torch.long), 4).float().to(dev)
        model.eval()
        with torch.no_grad():
            ops.set_gemm_precision("bf16x3"); o_ref, a_ref = model(img0, dl0, "0")
            ops.set_gemm_precision("bf16"); o_b, a_b = model(img0, dl0, "0")
        model.train()
        drift = {"what": "eval-mode logits, bf16 vs bf16x3 GEMMs, same weights and 2 images: max |diff| / max |logit|",
                 "out": round(float((o_b - o_ref).abs().max() / o_ref.abs().max()), 5), "aux": round(float((a_b - a_ref).abs().max() / a_ref.abs().max()), 5)}

    def step_batches(b):
        if args.model == "transfuse":
            return transfuse_train_step(model, b, optimizer=opt, accumulator=accum, fuse_domains=args.fuse_images >= len(domains) * args.batch)
        if args.model != "base":
            return mdvit_train_step(model, b, optimizer=opt, accumulator=accum, merged_sweeps=not args.reference_sweeps, fuse_domains=fuse)
        return base_train_step(model, b, optimizer=opt, accumulator=accum)

    host_pool = None
    if args.host_inputs:
        # what a DataLoader hands over (create_dataset.py:119-189 before its float conversion): uint8 HWC images, uint8 masks
        host_pool = []
        for s_i in range(2):
            g = torch.Generator().manual_seed(4321 + rank + 1000 * s_i)
            host_pool.append([(torch.randint(0, 256, (args.batch, args.size, args.size, 3), generator=g, dtype=torch.uint8).pin_memory(),
                               (torch.rand((args.batch, 1, args.size, args.size), generator=g) < 0.2).to(torch.uint8).pin_memory(),
                               torch.full((args.batch,), d, dtype=torch.long)) for d in domains])

    def step(i):
        if host_pool is not None:
            b = [(ops.image_normalize_u8(u8.to(dev, non_blocking=True)), lab.to(dev, non_blocking=True).float(), sid)
                 for (u8, lab, sid) in host_pool[i % len(host_pool)]]
            return step_batches(b)
        b = pool[i % len(pool)]
        return graphed(b) if graphed is not None else step_batches(b)

    graphed = None
    if args.graph:
        from mdvit_amd.graph import GraphedStep
        # --graph was asked for: a capture failure is an error (non-zero exit), never a silent eager run
        graphed = GraphedStep(lambda b: step_batches(b), pool[0], warmup=2, fuse_domains=fuse)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    inflight = []
    waited = [0.0]                      # seconds the host spent in throttle() during the timed steps (not enqueue work)

    def throttle():                     # called before a step is enqueued; done() after it
        while len(inflight) >= max(1, args.max_inflight):
            tw = time.perf_counter()
            inflight.pop(0).synchronize()
            waited[0] += time.perf_counter() - tw

    def done():
        e = torch.cuda.Event()
        e.record()
        inflight.append(e)

    # Settling, before the W warm-up steps.  On a FRESH box (the container image is still paging in) the first process runs its host side up to 1.8x
    # slower for a minute or two -- 39 ms of enqueue work per step instead of 22, which makes the 36.5 ms step host-bound (measured: the first bench process
    # on a box 395 images/s, the second 433, the third and later 435-438).  The settle loop repeats untimed steps, five at a time, until the host's enqueue
    # time sits comfortably under the step time (the GPU is the limit again), for at most --settle-seconds; what it did is reported in the JSON (`settle`).
    settle = {"steps": 0, "seconds": 0.0, "host_ms_first": None, "host_ms_last": None}
    if args.settle_seconds > 0 and not args.graph and world == 1 and args.model in ("mdvit", "mdvit_dsn") and args.decoder != "Transformer":
        # (the configurations whose step is GPU-bound on a warm box; TransFuse / BASE / the Transformer peers are host-bound by nature and would only burn the budget)
        ts0 = time.perf_counter()
        hist = []
        while time.perf_counter() - ts0 < args.settle_seconds:
            waited[0] = 0.0
            th = time.perf_counter()
            for j in range(5):
                throttle()
                step(settle["steps"] + j)
                done()
            host_ms = 1e3 * (time.perf_counter() - th - waited[0]) / 5
            torch.cuda.synchronize()
            step_ms = 1e3 * (time.perf_counter() - th) / 5
            inflight.clear()
            settle["steps"] += 5
            hist.append(host_ms)
            # settled = the GPU is the limit with margin AND the host has stopped getting faster (a box that is still paging the image in keeps improving for
            # a minute: stopping at the first window under the margin measured 458 images/s with 26.8 ms of enqueue work on a box that does 470+ once warm)
            stable = len(hist) >= 3 and hist[-1] > 0.97 * hist[-2] and hist[-2] > 0.97 * hist[-3]
            if stable and (host_ms < 0.75 * step_ms or time.perf_counter() - ts0 > 10.0):
                break
        settle.update(seconds=round(time.perf_counter() - ts0, 1), host_ms_first=round(hist[0], 1), host_ms_last=round(hist[-1], 1))
        waited[0] = 0.0
    use_events = not args.no_kernel_events and not args.graph
    # Round 6: the roofline kernel is found and timed by the LIBRARY's launch sampler (mdvit_gemm_sampler): kernel begin / end timestamps around the launches mdvit_gemm_f32
    # issues wherever it is called from -- the C-level block entry's products on the weight-gradient stream included, which the Python wrapper's event sampler never saw
    # (the largest kernel of the traced step was not a candidate).  --by-shape / --detail keep the wrapper's per-shape table.
    lib_sampler = use_events and not (args.by_shape or args.detail)
    import ctypes as _C
    from mdvit_amd import _lib as _L

    def sampler_read():
        out, i = {}, 0
        nm, seen, timed, ms = _C.create_string_buffer(160), _C.c_int64(), _C.c_int64(), _C.c_double()
        while _L.load().mdvit_gemm_sampler_read(i, nm, 160, _C.byref(seen), _C.byref(timed), _C.byref(ms)) == 0:
            out[nm.value.decode()] = {"n": int(timed.value), "ms": float(ms.value), "launches": int(seen.value), "flop": 0.0, "bytes": 0.0, "event_pair_overhead_ms": 0.0,
                                      "timer": "kernel begin/end timestamps (hipExtLaunchKernelGGL start/stop events) from the library's launch sampler: every launch site"}
            i += 1
        return out

    dominant, dom_stride = None, 1
    for i in range(args.warmup):
        throttle()
        scout = use_events and not (args.by_shape or args.detail) and i == args.warmup - 1
        if scout:                       # the last warm-up step times EVERY launch of mdvit_gemm_f32 to find the dominant kernel ...
            _L.call("mdvit_gemm_sampler", None, 1)
        step(i)
        if scout:
            t = {k: v for k, v in sampler_read().items() if v["n"] > 0}
            if t:
                dominant, drec = max(t.items(), key=lambda kv: kv[1]["ms"] * kv[1]["launches"] / kv[1]["n"])
                dom_stride = max(1, drec["launches"] // 32)       # ~32 timed launches per step: the events must not become the host's load
        done()
    fence()
    inflight.clear()
    waited[0] = 0.0
    if lib_sampler and dominant is not None:      # ... the timed steps time a sample of THAT kernel's launches only
        _L.call("mdvit_gemm_sampler", dominant.encode(), dom_stride)
    elif use_events:
        ops.kernel_events_begin(by_shape=args.by_shape, only=dominant, stride=dom_stride)
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        throttle()
        last = step(args.warmup + i)
        done()
    t_enq = time.perf_counter() - t0          # the host has ENQUEUED every step (the GPU is still running them unless the host is the limit)
    fence()
    dt = time.perf_counter() - t0
    if lib_sampler and dominant is not None:
        table = {k: v for k, v in sampler_read().items() if v["n"] > 0}
    else:
        table = ops.kernel_events_end() if use_events else {}
    # one more, untimed step under the library's GEMM launch ledger: EVERY mdvit_gemm_f32 call of a step by kernel symbol -- the C-level block entry's included, which the
    # event sampler above never sees -- so that the roofline line's launch count / bytes per launch and the committed rocprofv3 average describe one launch population
    ledger = {}
    if use_events and table:
        _L.call("mdvit_gemm_ledger", 1)
        step(args.warmup + args.steps)
        fence()
        _L.call("mdvit_gemm_ledger", 0)
        nm, nl, fl, by = _C.create_string_buffer(160), _C.c_int64(), _C.c_double(), _C.c_double()
        i = 0
        while _L.load().mdvit_gemm_ledger_read(i, nm, 160, _C.byref(nl), _C.byref(fl), _C.byref(by)) == 0:
            ledger[nm.value.decode()] = {"launches": int(nl.value), "flop": float(fl.value), "bytes": float(by.value)}
            i += 1
        if lib_sampler:          # the sampler's records carry no shapes: the sample's flops / bytes = the symbol's per-launch average over ALL its launches of a step x launches timed
            for k, v in table.items():
                led_k = ledger.get(k)
                if led_k and led_k["launches"]:
                    v["flop"], v["bytes"] = led_k["flop"] / led_k["launches"] * v["n"], led_k["bytes"] / led_k["launches"] * v["n"]
        shapes_out = os.environ.get("MDVIT_BENCH_GEMM_SHAPES")
        if shapes_out:          # measurement aid: one more untimed step with the ledger keyed by (kernel, shape, epilogue operands) -> a text table
            # EVERY rank runs the extra step (it issues the bucket all-reduces and fence() holds a barrier: ADVICE r05); rank 0 writes the file
            _L.call("mdvit_gemm_ledger", 2)
            step(args.warmup + args.steps + 1)
            fence()
            _L.call("mdvit_gemm_ledger", 0)
            rows, i = [], 0
            nm2 = _C.create_string_buffer(256)
            while _L.load().mdvit_gemm_ledger_read(i, nm2, 256, _C.byref(nl), _C.byref(fl), _C.byref(by)) == 0:
                rows.append((nm2.value.decode(), int(nl.value), float(fl.value), float(by.value)))
                i += 1
            if rank == 0:
                with open(shapes_out, "w") as f:
                    for r in sorted(rows, key=lambda r: -r[3]):
                        f.write(f"{r[1]:4d} x {r[3] / r[1] / 1e6:9.2f} MB {r[2] / r[1] / 1e9:9.2f} GF  {r[0]}\n")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_buckets, overlapped_buckets = len(accum.reducer.buckets), accum.overlapped_buckets
    rccl_ranks, devices = 1, [torch.cuda.get_device_name(dev) + f" (cuda:{local_rank})"]
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                       # what RCCL itself sees: one contribution per rank
        rccl_ranks = int(ones.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    imgs_per_step = len(domains) * args.batch * world
    value = imgs_per_step * args.steps / dt
    loss_val = float(last["loss"]) if last is not None else float("nan")

    # fwd / bwd / optimizer split (the metric's "fwd+bwd ms"): two extra, untimed, instrumented steps -- each phase ends with an
    # event on the main stream (the backward's events come after the side stream has been joined)
    phase_ms = None
    if args.model in ("mdvit", "mdvit_dsn") and not args.graph:
        acc_ms = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
        reps = 2
        for i in range(reps):
            evs = []
            mdvit_train_step(model, pool[i % len(pool)], optimizer=opt, accumulator=accum, merged_sweeps=not args.reference_sweeps,
                             fuse_domains=fuse, phase_events=evs)
            torch.cuda.synchronize()
            for (_, e0), (tag, e1) in zip(evs[:-1], evs[1:]):
                acc_ms[tag] += e0.elapsed_time(e1)
        phase_ms = {"fwd_incl_losses": round(acc_ms["fwd"] / reps, 3), "bwd_two_sweeps_incl_grad_accumulation": round(acc_ms["bwd"] / reps, 3),
                    "optimizer": round(acc_ms["opt"] / reps, 3)}

    if rank == 0:
        roof = None
        if table:
            name, rec = max(table.items(), key=lambda kv: kv[1]["ms"])
            ovh_ms = rec.get("event_pair_overhead_ms", 0.0)          # an empty event pair, measured after the timed region
            secs = max(rec["ms"] - rec["n"] * ovh_ms, 0.5 * rec["ms"]) * 1e-3
            tf, gbs = rec["flop"] / secs / 1e12, rec["bytes"] / secs / 1e9
            # the dominant GEMM variant is priced against BOTH roofs; "bound" is the one that is closer
            frac_mfma, frac_hbm = tf / peak_mfma, gbs / PEAK_HBM_GBS
            pmc = {}
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f).get("kernels", {})
            except Exception:
                pass
            default_leg = args.model == "mdvit" and args.batch == 4 and args.size == 512        # the command the committed --pmc passes ran
            traffic = pmc.get(name.split("+")[0], {}).get("hbm_bytes_per_launch") if default_leg else None
            if frac_hbm > frac_mfma:
                roof = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4)}
            else:
                roof = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": round(peak_mfma, 1), "unit": "TFLOP/s", "frac": round(frac_mfma, 4)}
            launches = rec.get("launches", rec["n"])
            led = ledger.get(name.split("+")[0].split(" M=")[0])
            rp_us = _rocprof_avg_us(name, args.model) if (args.model != "mdvit" or default_leg) else None
            if rp_us and led:
                # rocprofv3's average runs over ALL launches of this symbol in a step; so do the ledger's bytes / flops per launch (round 4 divided the event SAMPLE's
                # bytes by it: two populations -- the sample never holds the block entry's launches)
                rp_gbs, rp_tf = led["bytes"] / led["launches"] / (rp_us * 1e-6) / 1e9, led["flop"] / led["launches"] / (rp_us * 1e-6) / 1e12
                roof["frac_rocprof"] = round(rp_gbs / PEAK_HBM_GBS if roof["bound"] == "hbm" else rp_tf / peak_mfma, 4)
                roof["frac_live"] = roof["frac"]
            if led:
                roof["all_launches_per_step"] = led["launches"]
                roof["algorithmic_bytes_per_launch_all"] = round(led["bytes"] / led["launches"])
                roof["flop_per_launch_all"] = round(led["flop"] / led["launches"])
            big = _trace_largest(args.model) if (args.model != "mdvit" or default_leg) else None
            if big:
                roof["largest_kernel_of_committed_trace"] = big          # over ALL kernel families (the live sampler covers the mdvit_gemm_f32 entry: GEMMs, implicit convolutions, weight gradients)
            roof["traffic_source"] = ("profiles/pmc_traffic.json: HBM bytes per launch of this kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command (FETCH x2, gfx950) -- read from the file, NOT measured in this run") if default_leg else \
                "none: the committed --pmc passes ran the default MDViT bs=4 command, not this leg"
            roof.update({"traffic": traffic, "launches": launches, "timed_launches": rec["n"], "avg_launch_us": round(secs * 1e6 / rec["n"], 2),
                         "avg_launch_us_raw_events": round(rec["ms"] * 1e3 / rec["n"], 2), "event_pair_overhead_us": round(ovh_ms * 1e3, 2),
                         "timer": rec.get("timer"),
                         "rocprof_avg_launch_us": rp_us,
                         "flop_per_launch": round(rec["flop"] / rec["n"]), "algorithmic_bytes_per_launch": round(rec["bytes"] / rec["n"]),
                         "mfma_tflops": round(tf, 2), "hbm_gbs": round(gbs, 1),
                         # the symbol's share of the step: its live average x EVERY launch of it in a step (the ledger's count; the event-visible count without one)
                         "share_of_step": round(secs / rec["n"] * (led["launches"] if led else launches / args.steps) / (dt / args.steps), 4)})
            if args.detail:
                with open(args.detail, "w") as f:
                    json.dump({"step_ms": dt * 1e3 / args.steps, "kernels": table}, f, indent=1)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline(args.size)
            except Exception as e:           # the baseline is a report, never a reason to lose the GPU number
                cpu = {"error": repr(e)}
        extra = {}
        if world == 1 and not args.no_extra_legs and args.model == "mdvit" and args.batch != 32 and args.size == 512 and not args.host_inputs:
            # the metric is quoted "at bs=4/32" and the target on the MHSA+DA block at bs=32: run both as CHILD processes once
            # this process has handed its HBM back (a 128-image forward keeps ~115 GB of activations)
            del model, accum, opt, pool, last
            graphed = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            common = ["--precision", args.precision] + (["--no-side-stream"] if args.no_side_stream else [])
            b32 = _child_json([sys.executable, os.path.abspath(__file__), "--batch", "32", "--steps", "8", "--warmup", "3", "--no-cpu-baseline",
                               "--no-extra-legs", "--no-kernel-events"] + common, 600)
            extra["bs32"] = {k: b32.get(k) for k in ("value", "unit", "ms_per_step", "phase_ms", "steps", "warmup", "memory", "host_enqueue_ms_per_step", "error") if k in b32}
            if "config" in b32:
                extra["bs32"]["workload"] = b32["config"]["workload"]
            if args.precision == "bf16x3":
                # the bf16 speed mode NEXT TO the parity-mode headline (never instead of it)
                sp = _child_json([sys.executable, os.path.abspath(__file__), "--precision", "bf16", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                  "--no-cpu-baseline", "--no-extra-legs", "--no-kernel-events"] + (["--no-side-stream"] if args.no_side_stream else []), 600)
                extra["bf16_speed_mode"] = {k: sp.get(k) for k in ("value", "unit", "ms_per_step", "dtype", "drift_vs_parity_mode", "error") if k in sp}
            # BASELINE configs[0] (BASE, 1 domain x bs=4, 512x512: the case cpu_baseline.base_bs4 times on the host) on the GPU
            # (4 images per step: the eager step is host-bound at ~17 ms, so this leg replays the captured HIP graph of the step)
            bb = _child_json([sys.executable, os.path.abspath(__file__), "--model", "base", "--batch", "4", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                              "--no-extra-legs", "--graph"] + common, 300)
            extra["base_bs4_gpu"] = {k: bb.get(k) for k in ("value", "unit", "ms_per_step", "error") if k in bb}
            extra["base_bs4_gpu"]["how"] = "whole-step HIP graph replay (eager: 238 images/s, host-bound)"
            if "config" in bb:
                extra["base_bs4_gpu"]["workload"] = bb["config"]["workload"]
            # BASELINE configs[3]: bs=16 per domain, bf16 GEMMs / fp32 losses (one 64-image fused forward)
            b16 = _child_json([sys.executable, os.path.abspath(__file__), "--batch", "16", "--precision", "bf16", "--steps", "5", "--warmup", "3", "--no-cpu-baseline",
                               "--no-extra-legs", "--no-kernel-events"] + (["--no-side-stream"] if args.no_side_stream else []), 600)
            extra["bs16_bf16"] = {k: b16.get(k) for k in ("value", "unit", "ms_per_step", "dtype", "phase_ms", "drift_vs_parity_mode", "memory", "error") if k in b16}
            if "config" in b16:
                extra["bs16_bf16"]["workload"] = b16["config"]["workload"]
            # BASELINE configs[4]: TransFuse_S_adapt, bs=8 per domain at its only legal size 256x256 (one 32-image fused step)
            tfl = _child_json([sys.executable, os.path.abspath(__file__), "--model", "transfuse", "--batch", "8", "--size", "256", "--steps", "10", "--warmup", "4",
                               "--no-cpu-baseline", "--no-extra-legs"] + common, 600)
            extra["transfuse_bs8"] = {k: tfl.get(k) for k in ("value", "unit", "ms_per_step", "host_enqueue_ms_per_step", "roofline", "error") if k in tfl}
            if "config" in tfl:
                extra["transfuse_bs8"]["workload"] = tfl["config"]["workload"]
            blk_json = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mdvit_block_roofline_{os.getpid()}.json")
            _child_json([sys.executable, os.path.join(ROOT, "tools", "block_roofline.py"), "--batch", "32", "--precision", args.precision,
                         "--json", blk_json], 600)
            try:
                with open(blk_json) as f:
                    blk = json.load(f)
                extra["block_bs32"] = {
                    "what": "one SerialBlock_adapt (MHSA + Domain Adapter + MLP) forward + backward at bs=32, 512x512, per encoder stage; "
                            "frac = STRICT operator-sum roofline bound (the MLP fused: no T x hidden bytes -- x, res -> y; gm, x -> dx; gm, x -> dW) / measured "
                            "(tools/block_roofline.py); frac_unfused = the same operator list with the MLP as two GEMMs per pass (rounds 1-3 quoted that one as frac); "
                            "frac_survey = SURVEY 8(d) whole-block fused-bf16 bound / measured",
                    "stages": [{"stage": r["stage"], "C": r["C"], "rows": r["rows"], "fwd_ms": round(r["fwd_ms"], 3), "bwd_ms": round(r["bwd_ms"], 3),
                                "frac": round(r.get("frac_strict", 0.0), 4), "frac_unfused": round(r["frac"], 4), "frac_survey": round(r["frac_of_survey_fused_bf16_bound"], 4),
                                "achieved_TBps": round(r["achieved_TBps"], 3), "achieved_TFLOPs": round(r["achieved_TFLOPs"], 1)} for r in blk["stages"]],
                    "all_stages_frac": round(blk["all_stages"].get("frac_strict", 0.0), 4), "all_stages_frac_unfused": round(blk["all_stages"]["frac"], 4)}
                if "stage0_mlp_kernels" in blk and "largest" in blk["stage0_mlp_kernels"]:
                    big = blk["stage0_mlp_kernels"]["largest"]
                    extra["roofline_block_bs32"] = {
                        "what": "the largest kernel of the stage-0 block at bs=32 (524288 tokens), timed alone with events on its launch stream",
                        "kernel": big["kernel"], "us": round(big["us"], 1), "algorithmic_bytes": round(big["algorithmic_bytes"]), "algorithmic_flop": round(big["algorithmic_flop"]),
                        "bound": big["bound"], "achieved_GBps": round(big["hbm_GBps"], 1), "frac_hbm": round(big["frac_hbm"], 4),
                        "achieved_useful_TFLOPs": round(big["useful_TFLOPs"], 1), "frac_mfma_bf16x3": round(big["frac_mfma_bf16x3"], 4),
                        "all_mlp_kernels": [{"kernel": r_["kernel"], "us": round(r_["us"], 1), "frac_hbm": round(r_["frac_hbm"], 4), "frac_mfma_bf16x3": round(r_["frac_mfma_bf16x3"], 4)}
                                            for r_ in blk["stage0_mlp_kernels"]["kernels"]],
                        # the other half of the named block: the attention core's two passes at the same shape (tools/block_roofline.py: stage0_attention_core)
                        "attention_core": [{"kernel": r_["kernel"], "us": round(r_["us"], 1), "algorithmic_bytes": round(r_["algorithmic_bytes"]), "frac_hbm": round(r_["frac_hbm"], 4)}
                                           for r_ in blk.get("stage0_attention_core", {}).get("passes", [])]}
                os.remove(blk_json)
            except Exception as e:
                extra["block_bs32"] = {"error": repr(e)}
            if args.precision == "bf16x3":
                # the same block in the mode BASELINE configs[1] / [3] name ("bf16" / "mixed bf16 / fp32 loss"): the bf16 speed mode's kernels against the bound of bf16 STORAGE
                # (2 bytes per element) and the dense bf16 MFMA roof (2.5 PFLOP/s) -- VERDICT r04 item 4
                _child_json([sys.executable, os.path.join(ROOT, "tools", "block_roofline.py"), "--batch", "32", "--precision", "bf16", "--json", blk_json], 600)
                try:
                    with open(blk_json) as f:
                        blk = json.load(f)
                    extra["block_bs32_bf16"] = {
                        "what": "block_bs32 in the bf16 speed mode (one bf16 plane per GEMM operand, h / du of the C = 128 MLP stored bf16, everything else fp32 in HBM) against the "
                                "STRICT operator-sum bound priced for bf16 storage (2 bytes per element) and the 2.5 PFLOP/s dense bf16 roof",
                        "stages": [{"stage": r["stage"], "C": r["C"], "fwd_ms": round(r["fwd_ms"], 3), "bwd_ms": round(r["bwd_ms"], 3), "frac": round(r.get("frac_strict", 0.0), 4),
                                    "frac_unfused": round(r["frac"], 4)} for r in blk["stages"]],
                        "all_stages_frac": round(blk["all_stages"].get("frac_strict", 0.0), 4)}
                    os.remove(blk_json)
                except Exception as e:
                    extra["block_bs32_bf16"] = {"error": repr(e)}
        line = {
            "metric": {"mdvit": "512x512 images/sec MDViT train step (fwd+bwd, two-sweep, AdamW)", "mdvit_dsn": "512x512 images/sec MDViT_DSN train step",
                       "base": "512x512 images/sec BASE train step", "transfuse": "256x256 images/sec TransFuse_S_adapt train step (fwd+bwd, structure_loss, AdamW)"}[args.model],
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (fp32 storage; GEMM operands split hi+lo bf16, fp32 accumulate)",
                      "bf16": "mixed bf16 / fp32 (C >= 320 GEMMs: one bf16 plane per operand; C <= 128 blocks: bf16x3 register-chained kernels with the C = 128 MLP's [tokens, hidden] tensors h / du STORED as bf16; fp32 accumulate, norms, losses and everything else in HBM)"}[args.precision], "data": "synthetic" + (" (inputs cross PCIe inside the timed region)" if args.host_inputs else ""),
            "config": {"workload": f"{ {'mdvit': 'MDViT Sup+' + args.decoder, 'mdvit_dsn': 'MDViT_DSN Sup+' + args.decoder, 'base': 'BASE', 'transfuse': 'TransFuse_S_adapt'}[args.model] } train step, {len(domains)} domain(s) x bs={args.batch} per GPU, "
                                   f"{args.size}x{args.size}, drop_rate=0.1 drop_path=0.1, {args.precision} GEMMs, data-parallel x{world}",
                       "images_per_step": imgs_per_step, "algorithmic_gflop_per_image": flop_per_img / 1e9, "final_loss": round(loss_val, 4),
                       # (inside `config` so that it survives into the driver's `parsed` record: a host-bound box is visible at a glance -- VERDICT r05 item 9)
                       "host_enqueue_ms_per_step": round((t_enq - waited[0]) * 1e3 / args.steps, 3), "gpu_ms_per_step": round(dt * 1e3 / args.steps, 3)},
            "model_flops_util": round(value / world * flop_per_img / 1e12 / peak_mfma, 4),
            "host_enqueue_ms_per_step": round((t_enq - waited[0]) * 1e3 / args.steps, 3),
            "host_throttle_wait_ms_per_step": round(waited[0] * 1e3 / args.steps, 3),
            "phase_ms": phase_ms, "settle": settle, "roofline": roof, "cpu_baseline": cpu, "drift_vs_parity_mode": drift,
            "world": world, "rccl_ranks": rccl_ranks, "devices": devices,
            "allreduce": {"buckets": n_buckets, "issued_under_the_aux_sweep": overlapped_buckets},
            "memory": {"peak_allocated_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2), "peak_reserved_gib": round(torch.cuda.max_memory_reserved() / 2**30, 2),
                       "device_mallocs": int(torch.cuda.memory_stats().get("num_device_alloc", 0)), "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0))},
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
