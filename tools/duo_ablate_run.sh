for n in ${DUO_ABLS:-0 1 2 3 32}; do
  if [ $n = 0 ]; then unset AGRL_HIP_LIB; else export AGRL_HIP_LIB=$PWD/agrl.pytorch_amd/lib/libagrl_hip_duoabl$n.so; fi
  echo "== DUO_ABL=$n"; python tools/conv1x1_duo_bench.py 20 256 - res,pool1 2>&1 | grep -v amdgpu.ids
done
