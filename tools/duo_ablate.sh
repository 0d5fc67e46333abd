#!/bin/bash
# builds agrl.pytorch_amd/lib/libagrl_hip_duoabl<N>.so for each ablation mask N given (conv1x1_duo.hip with -DDUO_ABL=N, the other
# objects from the shipped build): profiling only, results are wrong by design.  usage: tools/duo_ablate.sh 1 2 4 ...
set -e
cd "$(dirname "$0")/../agrl.pytorch_amd/csrc"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGRL_LP_F16=1 -DDUO_ABL=$n -c conv1x1_duo.hip -o build/conv1x1_duo_abl$n.o
  objs=$(ls build/*.o | grep -v conv1x1_duo)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv1x1_duo_abl$n.o -o ../lib/libagrl_hip_duoabl$n.so
  rm -f build/conv1x1_duo_abl$n.o
done
