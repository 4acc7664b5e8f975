#!/bin/bash
# A/B run of the persistent tridiagonalisation variants built by tools/ab_build.py (on the GPU box):
#   bash tools/trdp_ab.sh base direct late1 ...      -> gpurun_out/trdp_ab.log
# TRDP_AB_SIZES="2100 2001 2001" selects other blocks; TRDP_AB_STAMPS=1 adds the phase stamps of every variant
mkdir -p gpurun_out
LOG=gpurun_out/trdp_ab.log
: > $LOG
for v in "$@"; do
  echo "==== $v" >> $LOG
  HELFEM_AMD_LIB=$PWD/helfem_amd/build/variants/libhelfem_amd_$v.so timeout -k 10 120 python3 tools/trdp_blocks.py $TRDP_AB_SIZES 2>&1 | grep -v amdgpu.ids >> $LOG || exit 1
  if [ -n "$TRDP_AB_STAMPS" ]; then
    HELFEM_AMD_LIB=$PWD/helfem_amd/build/variants/libhelfem_amd_$v.so HELFEM_TRDP_STAMPS=1 timeout -k 10 120 python3 tools/trdp_blocks.py $TRDP_AB_SIZES 2>&1 | grep "k_trdp stamps" | tail -1 >> $LOG || exit 1
  fi
done
cat $LOG
