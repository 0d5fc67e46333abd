#!/bin/bash
# Builds ablation variants of the seam kernel (csrc/bottleneck_seam.hip, -DSEAM_ABL=n) into gpurun_out-independent scratch libraries
# under agrl.pytorch_amd/lib/ablate/ (git-ignored with the rest of lib/): usage tools/seam_ablate.sh 1 2 4 6 8 ...
set -e
cd "$(dirname "$0")/../agrl.pytorch_amd/csrc"
mkdir -p ../lib/ablate build
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DAGRL_LP_F16=1 -DSEAM_ABL=$n -c bottleneck_seam.hip -o ../lib/ablate/seam_$n.o
  objs=$(ls build/*.o | grep -v bottleneck_seam.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/ablate/seam_$n.o -o ../lib/ablate/libagrl_hip_seam$n.so
  rm ../lib/ablate/seam_$n.o
done
