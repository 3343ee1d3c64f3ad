#!/bin/bash
# Diagnostic build: the two cascade translation units with -DPDWT_CASC_TRACE (in-kernel timeline, casc_dev.hpp), linked with the
# objects of the normal build into pdwt_amd/lib_trace/.  Use: PDWT_LIBDIR=$PWD/pdwt_amd/lib_trace python tools/casc_trace.py
set -e
# (the background compiles below do not trip set -e: check that every object is newer than its source before linking)
cd "$(dirname "$0")/.."
python -m pdwt_amd.build > /dev/null
mkdir -p pdwt_amd/lib_trace pdwt_amd/build/trace
for f in dwt_casc dwt_casc_invw dwt_casc_inv3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPDWT_CASC_TRACE -c pdwt_amd/csrc/$f.hip -o pdwt_amd/build/trace/$f.o &
done
wait
for f in dwt_casc dwt_casc_invw dwt_casc_inv3; do [ pdwt_amd/build/trace/$f.o -nt pdwt_amd/csrc/$f.hip ] || { echo "trace build of $f.hip FAILED"; exit 1; }; done
OBJS=$(ls pdwt_amd/build/*.o | grep -v "/dwt_casc.o\|/dwt_casc_invw.o\|/dwt_casc_inv3.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pdwt_amd/lib_trace/libpdwt_hip.so $OBJS pdwt_amd/build/trace/dwt_casc.o pdwt_amd/build/trace/dwt_casc_invw.o pdwt_amd/build/trace/dwt_casc_inv3.o
for v in "libpdwt.so:" "libpdwtd.so:-DDOUBLEPRECISION"; do
  n=${v%%:*}; fl=${v#*:}
  g++ -O2 -std=c++17 -fPIC -shared $fl -o pdwt_amd/lib_trace/$n pdwt_amd/csrc/wt.cpp pdwt_amd/csrc/wt_capi.cpp -Lpdwt_amd/lib_trace -lpdwt_hip '-Wl,-rpath,$ORIGIN'
done
ls -la pdwt_amd/lib_trace
