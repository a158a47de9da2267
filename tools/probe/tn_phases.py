"""Where a slab period of the weight-gradient kernel goes: shader-cycle stamps (s_memtime) of ONE thread of workgroup (tile 0, split 0) around
load issue | MFMA issue | split + LDS store | barrier, slab by slab.  Needs the variant build (gemm_tn.hip with -DMDVIT_TN_PHASES linked into
mdvit_amd/lib/variants/libmdvit_hip_tnphases.so):   MDVIT_HIP_LIB=mdvit_amd/lib/variants/libmdvit_hip_tnphases.so python tools/probe/tn_phases.py [M N K]"""
import ctypes, os, sys, torch
_r = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _r)
from mdvit_amd import _lib, ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1280, 320, 16384)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mdvit_tn_debug_buffer.argtypes = [ctypes.c_void_p]
A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")
dbg = torch.zeros(64 * 8, dtype=torch.int64, device="cuda")
_lib.load().mdvit_tn_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) if hasattr(_lib.load(), "mdvit_tn_debug_buffer") else lib.mdvit_tn_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3):
    ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=True, accumulate=True, precision=1)
torch.cuda.synchronize()
d = dbg.cpu().view(64, 8).tolist()
print(f"[{K} x {M}]^T [{K} x {N}]: cycles of thread 0, workgroup (tile 0, split 0)")
print("slab   load-issue   mfma-issue   (wait for the set's loads)   split+store   barrier    period")
prev4 = None
rows = [r for r in d if r[1] and r[4]]
for i, r in enumerate(rows[:40]):
    start = r[0] if r[0] else (prev4 if prev4 else r[1])
    print(f"{i:4d}   {r[1] - start:10d}   {r[2] - r[1]:10d}   {r[6] - r[5] if r[5] else -1:12d}            {r[3] - (r[6] if r[6] else r[2]):11d}   {r[4] - r[3]:7d}   {r[4] - start:7d}")
    prev4 = r[4]
if len(rows) > 4:
    body = rows[2:-1]
    import statistics as st
    def med(f): return st.median(f(a, b) for a, b in zip(body[1:], body[:-1]))
    print("median period over the body:", med(lambda a, b: a[4] - b[4]))
