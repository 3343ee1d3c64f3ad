for e in PDWT_CASC_SPEC=3 PDWT_CASC_SPEC=0 PDWT_CASC_SPEC=3 PDWT_CASC_SPEC=0; do env $e python bench.py --config c2 --steps 4000 --warmup 200 --cpu-seconds 0 --no-others --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['power'] or {}
print('$e', round(d['ms_per_step']*1e3,2), 'us  sclk', (p.get('sclk_mhz') or {}).get('mean'), 'MHz  power', (p.get('socket_w') or {}).get('mean'), 'W')"; done
