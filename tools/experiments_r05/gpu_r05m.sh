cd /tmp; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for sh in 104,64,128 208,32,64 52,128,256; do
echo "== $sh"; FSD_LB_ONLY=$sh python $R/tools/layer_bench.py all 2>&1 | grep -v "class_scale\|amdgpu"
echo "-- swapped (dgrad shape)"; FSD_LB_SWAP=1 FSD_LB_ONLY=$sh python $R/tools/layer_bench.py fwd 2>&1 | grep -v "class_scale\|amdgpu"
done
echo "== 104 direct (WINO4_MIN_CH=256)"; FSD_WINO4_MIN_CH=256 FSD_LB_ONLY=104,64,128 python $R/tools/layer_bench.py all 2>&1 | grep -v "class_scale\|amdgpu"
