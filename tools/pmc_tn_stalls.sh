#!/bin/bash
# Where the cycles of the weight-gradient kernel go:   bash tools/pmc_tn_stalls.sh [M N K]   -> gpurun_out/pmc_tn_stalls.txt
# (tools/pmc_stalls.sh with consistent units: round 3's version printed SQ_VALU_MFMA_BUSY_CYCLES -- cycles -- as a share of SQ_WAVE_CYCLES -- quad-cycles.)
exec bash "$(dirname "$0")/pmc_stalls.sh" gemm_tn pmc_tn_stalls.txt tools/probe/tn_one_shape.py "$@"
