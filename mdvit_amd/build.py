"""Build libmdvit_hip.so (gfx950) in-tree:  python -m mdvit_amd.build

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
snapshot (it is NOT gpurun-ignored)."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libmdvit_hip.so")
SOURCES = ["abi.hip", "gemm.hip", "gemm_tn.hip", "gemm_bp.hip", "gemm_ph.hip", "gemm_pm.hip", "mlp.hip", "mlp_rc.hip", "block.hip", "norm.hip", "conv.hip", "attn.hip", "loss.hip", "optim.hip", "transfuse.hip", "sdpa.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-file extras.  -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 adds / muls of an activation into v_pk_* instructions,
# which next to MFMAs cost ~20 cycles more than the two plain VALU they replace (MI355X_MICROARCH.md, per-instruction constants)
# The GEMM family (round 3): the operand split's subtractions came out as v_pk_add_f32 -- without the pass the weight-gradient kernel alone is 7-9 % faster
# ([65536 x 1024]^T [65536 x 128] 96.5 -> 87.8 us), bs=32 515-521 -> 527-529 images/s, bs=4 +0.3 %.  attn.hip / sdpa.hip / mlp.hip measured too: no change / -0.4 %
# on TransFuse, so they keep the default.  `-mllvm -amdgpu-sched-strategy=max-ilp` on gemm.hip / gemm_tn.hip / mlp_rc.hip: no change either (step and block_bs32
# within noise), not used.
EXTRA_FLAGS = {"mlp_rc.hip": ["-fno-slp-vectorize"], "gemm.hip": ["-fno-slp-vectorize"], "gemm_tn.hip": ["-fno-slp-vectorize"], "gemm_bp.hip": ["-fno-slp-vectorize"], "gemm_ph.hip": ["-fno-slp-vectorize"], "gemm_pm.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    stamp = obj + ".sha"
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_tile.h"), os.path.join(CSRC, "gemm_body.inc"), os.path.join(CSRC, "gemm_bp.h"),
            os.path.join(HERE, "..", "include", "mdvit_hip.h")]
    extra = EXTRA_FLAGS.get(src, [])
    dig = _digest(deps) + " ".join(extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
