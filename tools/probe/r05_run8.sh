cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "mlp or block or layernorm" 2>&1 | tail -4
python tools/block_roofline.py --batch 32 --stages 0 2>&1 | grep -v amdgpu | tail -12
python - <<'PY'
from mdvit_amd._lib import call
call("mdvit_block_config", 0)
import runpy, sys
sys.argv = ["block_roofline.py", "--batch", "32", "--stages", "0"]
runpy.run_path("tools/block_roofline.py", run_name="__main__")
PY
