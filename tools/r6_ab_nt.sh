#!/bin/bash
# interleaved A/B on one box: C2 headline (one image, Infinity-Cache-resident) with non-temporal loads in the cascade kernels
for r in 1 2 3 4; do
  for V in "" pdwt_amd/lib_c2ntf pdwt_amd/lib_c2nti pdwt_amd/lib_c2ntfi; do
    env ${V:+PDWT_LIBDIR=$PWD/$V} python bench.py --config c2 --steps 2000 --warmup 200 --cpu-seconds 0 --no-others 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s us/step %.2f  ' % ('$V'.replace('pdwt_amd/','') or 'lib', d['ms_per_step']*1e3), {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, 'rt %.1e' % d['roundtrip_max_rel_err'])"
  done
done
