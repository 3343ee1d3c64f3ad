#!/bin/bash
# A/B of library variants on ONE box (timings drift box to box): tools/ab_c2.sh <libdir_a> <libdir_b> [reps]  ("" = pdwt_amd/lib)
A=$1; B=$2; REPS=${3:-3}
run() {
  local V=$1
  env ${V:+PDWT_LIBDIR=$PWD/$V} python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-1000} --warmup 100 --cpu-seconds 0 --no-others 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s us/step %.2f  ' % ('$V' or 'lib', d['ms_per_step']*1e3), {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, 'rt %.1e' % d['roundtrip_max_rel_err'])"
}
for r in $(seq $REPS); do run "$A"; run "$B"; done
