cd $GRAFT_REPO_ROOT
for m in 0 1 2; do for t in 8 16 32 64; do echo "mode $m tiles $t"; python tools/attn_time.py --apply-mode $m --apply-tiles $t --stages 0,1 2>&1 | grep stage | cut -c1-40,118-200; done; done
for t in 4 8 16 32; do echo "apply3 tiles $t"; python tools/attn_time.py --apply-mode 0 --apply-tiles $t --stages 2,3 2>&1 | grep stage | cut -c1-40,118-200; done
