#!/bin/bash
# memory-side PMC passes (ONE counter per pass, each under a timeout: two derived counters in one pass abort rocprofv3 and hang its finalisation)
LIBV=${1:-lib}
export PDWT_LIBDIR=$PWD/pdwt_amd/$LIBV
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_MISS_sum"; do
  i=$((i+1))
  echo "== $LIBV exp2=$PDWT_EXP2: $CNT"
  timeout 150 bash tools/prof_pmc_cmd.sh mem_${LIBV}_${PDWT_EXP2}_$i "$CNT" python $PWD/tools/lat_time.py 1 2>&1 | grep -a "lat<\|f64lds" 
done
