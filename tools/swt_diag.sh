#!/bin/bash
# Diagnostic libraries for the fused SWT level kernels: pdwt_amd/lib_diag<k> = the product library with the three fused SWT translation
# units compiled with -DPDWT_SWTF_DIAG=<k> (1: stores folded onto 64 rows, 2: loads folded, 3: both -> results are WRONG, timings tell
# what the kernels cost without their memory traffic).  Run the bench against one with PDWT_LIBDIR=$PWD/pdwt_amd/lib_diag<k>.
cd "$(dirname "$0")/.."
SRCS=$(python -c "import pdwt_amd.build as b; print(' '.join(b.HIP_SOURCES))")
for K in ${@:-1 2 3}; do
  mkdir -p /tmp/swt_diag$K pdwt_amd/lib_diag$K
  for f in fwd inv invp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPDWT_SWTF_DIAG=$K -c pdwt_amd/csrc/swt_fused_$f.hip -o /tmp/swt_diag$K/swt_fused_$f.o &
  done
done
wait
for K in ${@:-1 2 3}; do
  objs=""
  for s in $SRCS; do n=${s%.*}; if [ -f /tmp/swt_diag$K/$n.o ]; then objs="$objs /tmp/swt_diag$K/$n.o"; else objs="$objs pdwt_amd/build/$n.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pdwt_amd/lib_diag$K/libpdwt_hip.so $objs
  cp pdwt_amd/lib/libpdwt.so pdwt_amd/lib/libpdwtd.so pdwt_amd/lib_diag$K/
done
ls -d pdwt_amd/lib_diag*
