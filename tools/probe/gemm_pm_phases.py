"""Where a K tile of gemm_pm.hip goes: shader-clock stamps of lane 0 of wave 0 (half X) and wave 4 (half Y) of workgroup 0.  Needs the variant build
    python tools/build_variant.py pmstamps gemm_pm.hip -DMDVIT_PM_STAMPS=1
    MDVIT_HIP_LIB=mdvit_amd/lib/variants/libmdvit_hip_pmstamps.so python tools/probe/gemm_pm_phases.py [M N K]
stamps per tile: 0 top of L | 1 fragment reads issued | 2 behind the vmcnt wait | 3 behind barrier 1 | 4 four MFMA groups + this tile's loads issued | 5 behind the wait for the A registers |
6 A converted + written, last MFMA group issued | 7 behind lgkmcnt(0) + barrier 2"""
import ctypes, os, sys, statistics as st
import torch
_r = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _r); sys.path.insert(0, os.path.join(_r, "tools"))
from mdvit_amd import _lib
from mdvit_amd._lib import call
from gemm_bp_check import planes_of, run_bp
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 320, 1280)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mdvit_pm_debug_buffer.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(2 * 64 * 8, dtype=torch.int64, device="cuda")
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; out = torch.empty(M, N, device="cuda"); wp = planes_of(w)
call("mdvit_gemm_planes_force_plan", 6 if N % 160 == 0 else 7, 0)
for _ in range(30):
    run_bp(x, wp, M, N, K, a_f32=True, C_out=out)
torch.cuda.synchronize()
lib.mdvit_pm_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
run_bp(x, wp, M, N, K, a_f32=True, C_out=out)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 64, 8).tolist()
nt = min(K // 32, 64)
print(f"{M} x {N} x {K}: {nt} K tiles; cycles between stamps, medians over tiles 4 .. {nt - 3}")
names = ["reads issued", "vmcnt wait (B)", "barrier 1", "4 MFMA groups + issue", "vmcnt wait (A)", "convert + write + 2 groups", "lgkmcnt + barrier 2", "-> next tile's top"]
for h, lab in ((0, "X (3 column blocks)"), (1, "Y (2 column blocks)")):
    rows = d[h][:nt]
    body = range(4, nt - 2)
    segs = [st.median(rows[t][k + 1] - rows[t][k] for t in body) for k in range(7)] + [st.median(rows[t + 1][0] - rows[t][7] for t in body)]
    per = st.median(rows[t + 1][0] - rows[t][0] for t in body)
    print(f"  half {lab}: period {per:.0f} cycles")
    for n_, v in zip(names, segs):
        print(f"      {n_:30s} {v:8.0f}")
print("  (X's 'barrier 1' waits for Y's multiply phase and the other way round: the two halves run one barrier apart)")
