#!/bin/bash
# builds agrl.pytorch_amd/lib/libagrl_hip_fat3trace.so: conv3x3_fat.hip with phase stamps (-DFAT_ABL=16), the other objects from the shipped build
set -e
cd "$(dirname "$0")/../agrl.pytorch_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGRL_LP_F16=1 -DFAT_ABL=16 -c conv3x3_fat.hip -o build/conv3x3_fat_trace.o
objs=$(ls build/*.o | grep -v "conv3x3_fat")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv3x3_fat_trace.o -o ../lib/libagrl_hip_fat3trace.so
rm -f build/conv3x3_fat_trace.o
