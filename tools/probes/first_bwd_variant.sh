# time first_bwd_kernel built with extra -D flags ($1...) against the tree's build (timing only)
cd $GRAFT_REPO_ROOT/fewshot_detection_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$GRAFT_REPO_ROOT/include "$@" -c first_bwd.hip -o /tmp/first_bwd_v.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC profile.o conv.o conv_halo.o conv_halo_h.o conv_first.o /tmp/first_bwd_v.o conv_bf16.o conv_bf16v2.o winograd.o wgrad.o wgrad_halo.o wgrad_halo_h.o elementwise.o backward_ew.o region_loss.o augment.o -o /tmp/lib_v.so
cd $GRAFT_REPO_ROOT
cp fewshot_detection_amd/libfsdet_hip.so /tmp/lib_orig.so
for v in orig v; do cp /tmp/lib_$v.so fewshot_detection_amd/libfsdet_hip.so; echo "$v $*"; timeout 120 python tools/probes/first_bwd_time.py ${SHAPE:-} 2>&1 | grep fused | grep -v unfused; [ $v = v ] && timeout 100 python -m pytest tests/test_gpu_first_bwd.py -x -q -m gpu 2>&1 | tail -1; done
cp /tmp/lib_orig.so fewshot_detection_amd/libfsdet_hip.so
