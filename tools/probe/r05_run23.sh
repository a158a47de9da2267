cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/host_cprofile.py --model transfuse --batch 8 --steps 5 2>&1 | grep -v amdgpu.ids > $O/host_cprofile_transfuse.txt
head -90 $O/host_cprofile_transfuse.txt | cut -c1-160
