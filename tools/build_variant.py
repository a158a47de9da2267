"""Build a variant of libmdvit_hip.so with extra compiler flags on ONE source (tuning experiments; the product build is mdvit_amd/build.py):
    python tools/build_variant.py NAME SOURCE.hip [-DFLAG=1 ...]      ->  mdvit_amd/lib/variants/libmdvit_hip_NAME.so
Use it with  MDVIT_HIP_LIB=mdvit_amd/lib/variants/libmdvit_hip_NAME.so  (mdvit_amd/_lib.py)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import build as B

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
vdir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, f"{os.path.splitext(src)[0]}_{name}.o")
cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(B.CSRC, src), "-o", obj]
subprocess.run(cmd, check=True)
objs = [obj if s == src else os.path.join(B.OBJ, os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
lib = os.path.join(vdir, f"libmdvit_hip_{name}.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
print(lib)
