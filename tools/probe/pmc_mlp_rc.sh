#!/bin/bash
# PMC stall breakdown of the mlp_rc kernels (run on the GPU box from the repo root):  bash tools/probe/pmc_mlp_rc.sh   -> gpurun_out/pmc_mlp_rc_stalls.txt
exec bash "$(dirname "$0")/../pmc_stalls.sh" mlp_rc pmc_mlp_rc_stalls.txt tools/probe/mlp_rc_pmc_run.py
