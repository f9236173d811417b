#!/bin/bash
# bf16 conv tile sweep (tuning aid): per-layer forward / data-gradient rates at B = 64 with each tile of
# conv_bf16_dma_kernel forced (FSD_CONV_H_TILE: 0 = 128x128, 1 = 256x256, 2 = 192x256, 3 = 256x128 on 8 waves, 6 = 192x128),
# then the automatic choice; FSD_CONV_H_ILV / FSD_CONV_H_RING / FSD_CONV_H_WIDE switch the DMA interleave, the counted-vmcnt
# ring and the 16-byte epilogue.  Results land in gpurun_out/r03c/layers.log.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03c"; mkdir -p "$O"; cd "$R"
for t in 0 1 2 3 6 auto; do
  echo "== tile $t" | tee -a "$O/layers.log"
  if [ "$t" = auto ]; then unset FSD_CONV_H_TILE; else export FSD_CONV_H_TILE=$t; fi
  FSD_LB_DTYPE=bf16 python tools/layer_bench.py fwd 2>&1 | grep -v "class_scale\|amdgpu.ids" | tee -a "$O/layers.log"
done
