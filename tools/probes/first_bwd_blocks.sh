cd $GRAFT_REPO_ROOT
for b in 2048 1280 2560 1024 1536 3840 5120; do echo "blocks $b"; FSD_FB_BLOCKS=$b timeout 120 python tools/probes/first_bwd_time.py 2>&1 | grep fused | grep -v unfused; done
