# which pipe bounds first_bwd_kernel: timing with the S3 MFMAs (-DFB_NO_S3) / all MFMAs (-DFB_NO_MFMA) compiled out.  The macros are NOT in the tree: apply tools/experiments_r06/first_bwd_ablate.patch first.
cd $GRAFT_REPO_ROOT/fewshot_detection_amd/csrc
for v in FB_NO_S3 FB_NO_MFMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$GRAFT_REPO_ROOT/include -D$v -c first_bwd.hip -o /tmp/first_bwd_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC profile.o conv.o conv_halo.o conv_halo_h.o conv_first.o /tmp/first_bwd_$v.o conv_bf16.o conv_bf16v2.o winograd.o wgrad.o wgrad_halo.o wgrad_halo_h.o elementwise.o backward_ew.o region_loss.o augment.o -o /tmp/lib_$v.so
done
cd $GRAFT_REPO_ROOT
cp fewshot_detection_amd/libfsdet_hip.so /tmp/lib_orig.so
for v in orig FB_NO_S3 FB_NO_MFMA; do cp /tmp/lib_$v.so fewshot_detection_amd/libfsdet_hip.so; echo $v; timeout 120 python tools/probes/first_bwd_time.py 2>&1 | grep fused | grep -v unfused; done
cp /tmp/lib_orig.so fewshot_detection_amd/libfsdet_hip.so
