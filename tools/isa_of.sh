#!/bin/bash
# device assembly of one translation unit (no GPU needed):  tools/isa_of.sh dwt_lat [extra hipcc flags]  -> /tmp/<tu>.s + an opcode summary per kernel
tu=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o /tmp/$tu.s "$(dirname "$0")/../pdwt_amd/csrc/$tu.hip" || exit 1
python3 - /tmp/$tu.s <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = collections.Counter(re.findall(r'^\s+([a-z_0-9]+)', body, re.M))
    keep = ('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_pk_fma_f32', 'v_readlane_b32', 'v_writelane_b32', 'ds_read_b128', 'ds_read_b64', 'ds_read2_b64', 'ds_write_b64', 'ds_write_b128', 'ds_write2_b64',
            's_waitcnt', 's_load_dwordx16', 's_load_dwordx8', 's_load_dwordx4', 's_load_dwordx2', 'v_mov_b32', 'v_mov_b64', 'v_pk_mov_b32', 's_barrier', 'global_load_dwordx4', 'global_load_dwordx2',
            'global_store_dwordx2', 'global_store_dwordx4', 'scratch_load_dword', 'scratch_store_dword', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 's_mov_b32', 's_mov_b64', 's_nop')
    print(name[:60], sum(ops.values()), {k: ops[k] for k in keep if ops[k]})
for m in re.finditer(r'\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|name):\s+(\S+)', txt):
    print(m.group(1), m.group(2))
PY
