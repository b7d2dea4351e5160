#!/bin/bash
# MSM latency for several lanes-per-bucket settings of k_msm_bucket_sum (compile-time MSM_LPB); leaves the default build behind
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-ilp"
for l in 4 2 1 8; do
  hipcc $F -DMSM_LPB=$l -c spartan2_amd/csrc/capi_group.hip -o spartan2_amd/lib/capi_group.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o spartan2_amd/lib/libspartan_hip.so spartan2_amd/lib/capi_core.o spartan2_amd/lib/capi_group.o spartan2_amd/lib/capi_nifs.o spartan2_amd/lib/capi_sparse.o
  echo "LPB=$l"; timeout 100 python tools/msm_latency.py
done
