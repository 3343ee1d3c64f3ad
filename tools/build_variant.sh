#!/bin/bash
# A variant build of ONE OR MORE translation units, linked with the objects of the normal build into pdwt_amd/lib_<name>/.
#   tools/build_variant.sh <name> <tu[,tu...] without .hip> <hipcc flags...>
#   e.g.  tools/build_variant.sh nt1 dwt1d_fused -DPDWT_1D_NT=1
#         tools/build_variant.sh ilp swt_fused_fwd,swt_fused_inv,swt_fused_invp '-DPDWT_SWTF_HLENS(X)=X(14)' -mllvm -amdgpu-sched-strategy=max-ilp
# Use:  PDWT_LIBDIR=$PWD/pdwt_amd/lib_<name> python bench.py --config c4      (the variant directories are git-ignored and travel with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; tus=$2; shift 2
[ -f pdwt_amd/build/common.o ] || [ -n "$(ls pdwt_amd/build/*.o 2>/dev/null)" ] || python -m pdwt_amd.build > /dev/null
mkdir -p pdwt_amd/lib_$name pdwt_amd/build/var_$name
EXCL=""
for tu in ${tus//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c pdwt_amd/csrc/$tu.hip -o pdwt_amd/build/var_$name/$tu.o &
  EXCL="$EXCL|/$tu.o"
done
wait
for tu in ${tus//,/ }; do [ pdwt_amd/build/var_$name/$tu.o -nt pdwt_amd/csrc/$tu.hip ] || { echo "variant build of $tu.hip FAILED"; exit 1; }; done
OBJS=$(ls pdwt_amd/build/*.o | grep -v -E "${EXCL#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pdwt_amd/lib_$name/libpdwt_hip.so $OBJS pdwt_amd/build/var_$name/*.o
for v in "libpdwt.so:" "libpdwtd.so:-DDOUBLEPRECISION"; do
  n=${v%%:*}; fl=${v#*:}
  g++ -O2 -std=c++17 -fPIC -shared $fl -o pdwt_amd/lib_$name/$n pdwt_amd/csrc/wt.cpp pdwt_amd/csrc/wt_capi.cpp -Lpdwt_amd/lib_$name -lpdwt_hip '-Wl,-rpath,$ORIGIN'
done
echo "built pdwt_amd/lib_$name"
