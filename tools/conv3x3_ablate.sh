#!/bin/bash
# Scratch variants of the four-wave 3x3 kernel (csrc/conv3x3_fat.hip) under agrl.pytorch_amd/lib/ablate/ (git-ignored):
# (FATSRC=conv1x1_fat for csrc/conv1x1_fat.hip) usage tools/conv3x3_ablate.sh <tag> "<hipcc -D flags>" [<tag> "<flags>" ...]; each is ISA-checked (no scratch, no compiler AGPR use)
set -e -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/agrl.pytorch_amd/csrc"
mkdir -p ../lib/ablate build
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  bash "$root/tools/seam_check_isa.sh" $flags | grep "LP_F16=1.*${FATSRC:-conv3x3_fat}" || { echo "ISA check failed for $tag"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DAGRL_LP_F16=1 $flags -c ${FATSRC:-conv3x3_fat}.hip -o ../lib/ablate/fat_$tag.o
  objs=$(ls build/*.o | grep -v ${FATSRC:-conv3x3_fat}.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/ablate/fat_$tag.o -o ../lib/ablate/libagrl_hip_fat_$tag.so
  rm ../lib/ablate/fat_$tag.o
done
