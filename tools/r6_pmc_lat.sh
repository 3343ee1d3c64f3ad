#!/bin/bash
# PMC passes over the level-1 lattice kernels (8192^2 f64 db20 L1) for the library in $1 (default lib)
LIBV=${1:-lib}
export PDWT_LIBDIR=$PWD/pdwt_amd/$LIBV
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  echo "== $LIBV: $CNT"
  bash tools/prof_pmc_cmd.sh lat_${LIBV}_$i "$CNT" python $PWD/tools/lat_time.py 1 2>&1 | grep -a "lat<" 
done
