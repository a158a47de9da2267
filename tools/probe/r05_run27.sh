cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
for s in "16384 320 1280" "16384 320 320" "8192 512 2048"; do MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/variants/libmdvit_hip_pmstamps.so python tools/probe/gemm_pm_phases.py $s 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05e/gemm_pm_phases.txt
