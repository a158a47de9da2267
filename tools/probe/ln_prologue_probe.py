import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mdvit_amd import ops
from mdvit_amd.ops import call, _p, _stream
torch.manual_seed(0)
for (M, K, N, G) in ((1024, 64, 192, 1), (5000, 64, 192, 1), (4096, 128, 384, 4), (300000, 64, 192, 1)):
    d = "cuda"
    x = torch.randn(M, K, device=d) * 2 + 0.3
    ga = (1 + 0.5 * torch.randn(G, K, device=d)).contiguous(); be = (0.1 * torch.randn(G, K, device=d)).contiguous()
    W = torch.randn(N, K, device=d) * K ** -0.5; b = torch.randn(N, device=d) * 0.1
    Wp = torch.empty(2, N, K, device=d, dtype=torch.bfloat16)
    call("mdvit_split_planes_t", _p(W), K, _p(Wp), K, N * K, N, K, 0, 2, _stream())
    cur0, mean0, rstd0, y0 = torch.empty(M, K, device=d), torch.empty(M, device=d), torch.empty(M, device=d), torch.empty(M, N, device=d)
    cur1, mean1, rstd1, y1 = torch.empty(M, K, device=d), torch.empty(M, device=d), torch.empty(M, device=d), torch.empty(M, N, device=d)
    call("mdvit_layernorm_fwd", _p(x), _p(ga), _p(be), _p(cur0), _p(mean0), _p(rstd0), M, K, G, 1e-6, _stream())
    call("mdvit_linear_rc", _p(cur0), K, _p(Wp), N * K, _p(b), _p(y0), N, M, N, K, 0.0, 0, 0, None, 1, None, 0, None, _stream())
    call("mdvit_linear_rc_ln", _p(x), _p(ga), _p(be), G, 1e-6, _p(mean1), _p(rstd1), _p(cur1), _p(Wp), N * K, _p(b), _p(y1), N, M, N, K, _stream())
    torch.cuda.synchronize()
    for n, a, c in (("mean", mean0, mean1), ("rstd", rstd0, rstd1), ("cur", cur0, cur1), ("y", y0, y1)):
        print(M, K, n, "equal" if torch.equal(a, c) else f"max diff {float((a - c).abs().max()):.3e} (rel {float((a - c).abs().max() / a.abs().max()):.1e})")
