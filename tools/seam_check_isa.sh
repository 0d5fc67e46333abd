#!/bin/bash
# The seam kernel (csrc/bottleneck_seam.hip) and the four-wave conv kernels (csrc/conv3x3_fat.hip, conv1x1_fat.hip, conv1x1_duo.hip)
# name their AGPRs in inline asm; hipcc must therefore never touch an AGPR itself and never spill: this compiles the files for both
# 16-bit types and fails if a kernel instance contains scratch accesses or any use of an AGPR (v_accvgpr_*, or an a-register operand) outside the asm blocks, if a
# file yields NO kernel instance to check (a renamed kernel must not pass vacuously), or if a register budget is exceeded
# (kernel:max = the allocation the occupancy the kernel is designed for allows). tools/ring_hazard_check.py then replays each kernel's
# vector-memory queue and fails if any instruction names a VGPR whose asm-issued load is still in flight (a v_mov / re-use of a ring
# register in front of the counted wait that covers it).
# usage: tools/seam_check_isa.sh [EXTRA hipcc flags]  (exit 0 = clean); HIPCC overrides the compiler as in csrc/Makefile
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$(dirname "$0")/../agrl.pytorch_amd/csrc"
src=$PWD
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
rc=0
for f in bottleneck_seam:bottleneck_seam_kernel:512 conv3x3_fat:conv3x3_fat_kernelILi1E:256 conv3x3_fat:conv3x3_fat_kernelILi2E:512 conv3x3_fat:conv3x3_half_kernel:256 conv1x1_fat:conv1x1_fat:512 conv1x1_duo:conv1x1_duo_kernel:256 conv1x1_duo:conv1x1_duo_persist_kernel:256; do
  IFS=: read -r file kern maxreg <<< "$f"
  for lp in 1 0; do
    (cd "$tmp" && "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGRL_LP_F16=$lp "$@" -I"$src" -c "$src/$file.hip" -o x.o -save-temps=obj 2>/dev/null)
    s=$(ls "$tmp"/*gfx950.s)
    found=0
    for k in $(grep -o "^_ZN12_GLOBAL__N_1[0-9]*${kern}[A-Za-z0-9_]*:" "$s" | tr -d ':'); do
      found=$((found + 1))
      body=$(awk "/^$k:/,/^\\.Lfunc_end/" "$s")   # the whole function (a kernel may hold several s_endpgm)
      nscr=$(echo "$body" | grep -c 'scratch_' || true)
      nacc=$(echo "$body" | awk '/#ASMSTART/{i=1} /#ASMEND/{i=0} { if(!i && /v_accvgpr|[ ,]a\[?[0-9]+[]: ,]/) n++ } END{print n+0}')
      vg=$(grep -A40 "^\s*.amdhsa_kernel $k" "$s" | grep -o 'amdhsa_next_free_vgpr [0-9]*' | head -1)
      echo "LP_F16=$lp $k: scratch ops $nscr, compiler v_accvgpr ops $nacc ($vg, budget $maxreg)"
      if [ "$nscr" != 0 ] || [ "$nacc" != 0 ]; then rc=1; fi
      if [ -n "$vg" ] && [ "${vg##* }" -gt "$maxreg" ]; then echo "  register budget exceeded"; rc=1; fi
    done
    python3 "$src/../../tools/ring_hazard_check.py" "$s" "${kern}" | sed "s/^/LP_F16=$lp /" || rc=1
    if [ "${PIPESTATUS[0]}" != 0 ]; then rc=1; fi
    if [ "$found" = 0 ]; then echo "LP_F16=$lp $file.hip: no kernel matching $kern found -- nothing was checked"; rc=1; fi
    rm -f "$tmp"/*
  done
done
exit $rc
