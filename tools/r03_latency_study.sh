#!/bin/bash
# Round 3: what a level of the block-cooperative point addition costs and why (tools/fb_stamps.hip built three ways). Build here (hipcc cross-compiles):
#   bash tools/r03_latency_study.sh build      then on the GPU box:      bash tools/r03_latency_study.sh run > gpurun_out/r03_latency_study.txt
F="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value -w -Ispartan2_amd/csrc"
if [ "$1" = build ]; then
  hipcc $F tools/fb_stamps.hip -o tools/fb_stamps
  hipcc $F -DSP_COOP_PREDICATED tools/fb_stamps.hip -o tools/fb_stamps_predicated
  hipcc $F -DSP_ROWWISE_BASE_PRODUCT tools/fb_stamps.hip -o tools/fb_stamps_rowwise
  exit 0
fi
echo "== products: column-accumulation base-field product (default build)"
tools/fb_stamps | grep -E "one wave, dep|sparse EXEC rep 0|chain profile mode . rep 1" | cut -c1-210
echo "== products: row-wise base-field product (-DSP_ROWWISE_BASE_PRODUCT)"
tools/fb_stamps_rowwise | grep -E "one wave, dep|sparse EXEC rep 0|chain profile mode . rep 1" | cut -c1-210
echo "== round-commitment walk (33 scalars, 16-bit windows, four tree levels), products on every lane of a wave that has work (default)"
tools/fb_stamps | grep "variant" | cut -c1-260
echo "== the same with the products predicated on the lanes that take part (-DSP_COOP_PREDICATED: the form of rounds 1-2)"
tools/fb_stamps_predicated | grep "variant" | cut -c1-260
