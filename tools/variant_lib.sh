#!/bin/bash
# Build pdwt_amd/lib_<name>: the product library with some translation units recompiled with extra flags (A/B experiments on the GPU box:
# PDWT_LIBDIR=$PWD/pdwt_amd/lib_<name>).   tools/variant_lib.sh <name> "<flags>" file1.hip [file2.hip ...]
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; shift 2
mkdir -p /tmp/var_$NAME pdwt_amd/lib_$NAME
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c pdwt_amd/csrc/$f -o /tmp/var_$NAME/${f%.*}.o &
done
wait
objs=""
for s in $(python -c "import pdwt_amd.build as b; print(' '.join(b.HIP_SOURCES))"); do n=${s%.*}; if [ -f /tmp/var_$NAME/$n.o ]; then objs="$objs /tmp/var_$NAME/$n.o"; else objs="$objs pdwt_amd/build/$n.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pdwt_amd/lib_$NAME/libpdwt_hip.so $objs && cp pdwt_amd/lib/libpdwt.so pdwt_amd/lib/libpdwtd.so pdwt_amd/lib_$NAME/ && echo built pdwt_amd/lib_$NAME
