#!/bin/bash
# The seam kernel (csrc/bottleneck_seam.hip) names its AGPRs in inline asm; hipcc must therefore never touch an AGPR itself and
# never spill: this compiles the file for both 16-bit types and fails if a bottleneck_seam_kernel instance contains scratch
# accesses or v_accvgpr_* instructions outside the asm blocks. usage: tools/seam_check_isa.sh  (exit 0 = clean)
set -e
cd "$(dirname "$0")/../agrl.pytorch_amd/csrc"
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
rc=0
for lp in 1 0; do
  (cd "$tmp" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGRL_LP_F16=$lp -I"$OLDPWD" -c "$OLDPWD/bottleneck_seam.hip" -o seam.o -save-temps=obj 2>/dev/null)
  s=$(ls "$tmp"/*gfx950.s)
  for k in $(grep -o '^_ZN12_GLOBAL__N_122bottleneck_seam_kernel[A-Za-z0-9_]*:' "$s" | tr -d ':'); do
    body=$(awk "/^$k:/,/s_endpgm/" "$s")
    nscr=$(echo "$body" | grep -c 'scratch_' || true)
    nacc=$(echo "$body" | awk '/#ASMSTART/{i=1} /#ASMEND/{i=0} { if(!i && /v_accvgpr/) n++ } END{print n+0}')
    echo "LP_F16=$lp $k: scratch ops $nscr, compiler v_accvgpr ops $nacc"
    if [ "$nscr" != 0 ] || [ "$nacc" != 0 ]; then rc=1; fi
  done
  rm -f "$tmp"/*
done
exit $rc
