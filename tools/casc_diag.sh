#!/bin/bash
# Diagnostic libraries for the streamed inverse cascade (C2; the forward cascade lost its folding hooks when its row addressing went
# incremental -- commit df608f3 still has them; the inverse folds through its row cursors): pdwt_amd/lib_cdiag<k> = the product library with dwt_casc_inv3.hip
# compiled with -DPDWT_CASC_DIAG=<k> (1: stored rows folded onto 32 rows, 2: loaded rows folded, 3: both -> results are WRONG, the timings
# say what the kernels cost without their memory traffic).  PDWT_LIBDIR=$PWD/pdwt_amd/lib_cdiag<k> python bench.py --config c2 ...
cd "$(dirname "$0")/.."
SRCS=$(python -c "import pdwt_amd.build as b; print(' '.join(b.HIP_SOURCES))")
for K in ${@:-1 2 3}; do
  mkdir -p /tmp/casc_diag$K pdwt_amd/lib_cdiag$K
  for f in dwt_casc_inv3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPDWT_CASC_DIAG=$K -c pdwt_amd/csrc/$f.hip -o /tmp/casc_diag$K/$f.o &
  done
done
wait
for K in ${@:-1 2 3}; do
  objs=""
  for s in $SRCS; do n=${s%.*}; if [ -f /tmp/casc_diag$K/$n.o ]; then objs="$objs /tmp/casc_diag$K/$n.o"; else objs="$objs pdwt_amd/build/$n.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pdwt_amd/lib_cdiag$K/libpdwt_hip.so $objs
  cp pdwt_amd/lib/libpdwt.so pdwt_amd/lib/libpdwtd.so pdwt_amd/lib_cdiag$K/
done
ls -d pdwt_amd/lib_cdiag*
