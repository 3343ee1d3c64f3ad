#!/bin/bash
# Regenerate every profile artefact under profiles/ at the current tree, for all BASELINE configs (run on the GPU box through
# gpurun):   PDWT_COMMIT=$(git rev-parse --short HEAD) tools/make_all_profiles.sh <tag> [configs...]
#   per config : <tag>_<cfg>_kernel_trace.md (rocprofv3 --kernel-trace --stats), <tag>_<cfg>_pmc_fetch_size.md / _pmc_write_size.md
#                (separate --pmc passes), <tag>_<cfg>_bench_line.json, and the entry of pmc_traffic.json (with the kernel-source
#                hash bench.py checks: roofline.traffic_stale)
#   c2 only    : <tag>_c2_pmc_sq.md (SQ busy / wait counters of the cascade kernels), <tag>_c2_timeline.md (in-kernel timestamps,
#                needs pdwt_amd/lib_trace from tools/build_trace.sh), <tag>_c2_isa_mix.md comes from tools/isa_mix.py (no GPU)
# Results land in gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
TAG=${1:-rXX}; shift
CFGS=${@:-c2 c3 c4 c5}
R=$PWD; export PYTHONPATH=$R
OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT
for CFG in $CFGS; do
  bash tools/make_profiles.sh $TAG $CFG 1 > $OUT/${CFG}.log 2>&1
done
# SQ counters (instruction mix, busy / wait cycles) of the C3 and C5 kernels as well (only C2 had them)
for CFG in $CFGS; do
  [ "$CFG" = "c3" ] || [ "$CFG" = "c5" ] || continue
  cd /tmp && export TMPDIR=/tmp
  for SET in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    N=$(echo $SET | cut -d' ' -f1)
    D=$R/gpurun_out/pmc_${TAG}_${CFG}_$N; rm -rf $D
    timeout 400 rocprofv3 --pmc $SET --kernel-trace -d $D -o t --output-format csv -- python $R/bench.py --config $CFG --steps 6 --warmup 2 --cpu-seconds 0 --no-roofline --no-others --settle-ms 0 > $D.log 2>&1
    python $R/tools/summarize_profile.py $D $OUT/${TAG}_${CFG}_pmc_sq_$N.md "${TAG}, config $CFG: rocprofv3 --pmc $SET --kernel-trace -- python bench.py --config $CFG --steps 6 --warmup 2 --no-others"
  done
  cat $OUT/${TAG}_${CFG}_pmc_sq_*.md > $OUT/${TAG}_${CFG}_pmc_sq.md 2>/dev/null; rm -f $OUT/${TAG}_${CFG}_pmc_sq_*.md
  cd $R
done
if echo "$CFGS" | grep -qw c2; then
  cd /tmp && export TMPDIR=/tmp
  # the same workload streaming through HBM: 16 images per step through the batched entry
  D=$R/gpurun_out/prof_${TAG}_c2_batch_kt; rm -rf $D
  timeout 400 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python $R/bench.py --config c2_batch --steps 40 --warmup 10 --cpu-seconds 0 --no-others > $D.log 2>&1
  grep '^{' $D.log | tail -1 > $OUT/${TAG}_c2_batch_bench_line.json
  python $R/tools/summarize_profile.py $D $OUT/${TAG}_c2_batch_kernel_trace.md "${TAG}, config c2_batch: rocprofv3 --kernel-trace --stats -- python bench.py --config c2_batch --steps 40 --warmup 10 --cpu-seconds 0 --no-others"
  for SET in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY"; do
    N=$(echo $SET | cut -d' ' -f1)
    D=$R/gpurun_out/pmc_${TAG}_c2_$N; rm -rf $D
    timeout 400 rocprofv3 --pmc $SET --kernel-trace -d $D -o t --output-format csv -- python $R/bench.py --config c2 --steps 10 --warmup 3 --cpu-seconds 0 --no-roofline --no-others --settle-ms 0 > $D.log 2>&1
    python $R/tools/summarize_profile.py $D $OUT/${TAG}_c2_pmc_sq_$N.md "${TAG}, config c2: rocprofv3 --pmc $SET --kernel-trace -- python bench.py --config c2 --steps 10 --warmup 3 --no-others"
  done
  cat $OUT/${TAG}_c2_pmc_sq_*.md > $OUT/${TAG}_c2_pmc_sq.md 2>/dev/null; rm -f $OUT/${TAG}_c2_pmc_sq_*.md
  cd $R
  if [ -d pdwt_amd/lib_trace ]; then
    PDWT_LIBDIR=$R/pdwt_amd/lib_trace python tools/casc_trace.py --md $OUT/${TAG}_c2_timeline.md > /dev/null 2>&1
  fi
fi
# (raw rocprofv3 output stays on the box: gpurun only brings back 64 MiB)
rm -rf $R/gpurun_out/pmc_${TAG}_* $R/gpurun_out/prof_${TAG}_*
cd $R; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; ls $OUT
