cd $GRAFT_REPO_ROOT
python - <<'PY'
import itertools, torch, sys
sys.path.insert(0, '.')
from mdvit_amd import ops
torch.manual_seed(0)
C, Hd = 64, 512
def run(mode, M, drop):
    ops._mlp_rc_bwd = mode
    ops._key_counter = itertools.count(5)
    g_ = torch.Generator(device="cpu").manual_seed(3)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g_) * sc).cuda()
    x, res, g = r(M, C).requires_grad_(True), r(M, C), r(M, C)
    W1, b1, W2, b2 = r(Hd, C, sc=C ** -0.5).requires_grad_(True), r(Hd, sc=0.1).requires_grad_(True), r(C, Hd, sc=Hd ** -0.5).requires_grad_(True), r(C, sc=0.1).requires_grad_(True)
    rs = (torch.rand(4, generator=g_) < 0.9).float().cuda() / 0.9
    y = ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=rs if drop > 0 else None, drop_p=drop, rows_per_scale=(M + 3) // 4)
    y.backward(g)
    torch.cuda.synchronize()
    return [t.grad.clone() for t in (x, W1, b1, W2, b2)]
for M in (37, 4173, 70000):
    for drop in (0.0, 0.1):
        a, b = run("0", M, drop), run("1", M, drop)
        print(M, drop, [("same" if torch.equal(u, v) else f"{float((u - v).abs().max() / v.abs().max()):.2e}") for u, v in zip(a, b)])
PY
python tools/mlp_rc_time.py --tokens128 0 --rounds 2 2>&1 | grep -v amdgpu
