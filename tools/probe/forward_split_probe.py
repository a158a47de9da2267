"""How much of the bs=4 step's forward (one 16-image domain-batched pass, 9.2 ms alone on the GPU while the weight-gradient and aux-sweep streams idle) would two
concurrent 8-image passes (two domains each, BatchNorm statistics are per domain batch either way) on two streams recover?  Forward only, no autograd:
    python tools/probe/forward_split_probe.py
Prints ms per 16 images for: one fused pass | two 2-domain passes back to back on one stream | the same two passes on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mdvit_amd                                     # noqa: E402
from mdvit_amd import ops                            # noqa: E402
from mdvit_amd.synthetic import make_step_batches    # noqa: E402
from mdvit_amd.train import _fuse_batches            # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM").to(dev).train()
batches = make_step_batches(4, 512, rank=0, step=0, device=dev)
f4 = _fuse_batches(batches, 4, 4, True)
f2 = _fuse_batches(batches, 2, 4, True)
print("fused batches:", len(f4), "of", f4[0][0].shape[0], "images;", len(f2), "of", f2[0][0].shape[0], flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd(b):
    img, set_id, dl, G = b[0], b[2], b[3], b[4]
    Bd = img.shape[0] // G
    return model(img, dl, [str(int(set_id[g * Bd])) for g in range(G)])


def one():
    fwd(f4[0])


def two_serial():
    fwd(f2[0]); fwd(f2[1])


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        fwd(f2[0])
    with torch.cuda.stream(s2):
        fwd(f2[1])
    cur.wait_stream(s1); cur.wait_stream(s2)


with torch.no_grad():
    variants = (("one 16-image pass", one),) if os.environ.get("MDVIT_FWD_ONLY_ONE") else (("one 16-image pass", one), ("two 8-image passes, one stream", two_serial), ("two 8-image passes, two streams", two_streams))
    for name, fn in variants * 2:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        th = (time.perf_counter() - t0) / 10 * 1e3
        torch.cuda.synchronize()
        print(f"{name:34s}: {e0.elapsed_time(e1) / 10:6.2f} ms per 16 images (host enqueue {th:5.2f} ms)", flush=True)
