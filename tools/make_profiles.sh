#!/bin/bash
# Produce the profile summaries committed under profiles/ (run on the GPU box through gpurun):
#   tools/make_profiles.sh <tag> [config]   -> gpurun_out/profiles_<tag>/{*_kernel_trace.md,*_pmc_fetch.md,*_pmc_write.md,pmc_traffic.json}
# 1. rocprofv3 --kernel-trace --stats of `python bench.py --steps 500 --warmup 100` (no counters)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes
TAG=${1:-rXX}; CFG=${2:-c2}; PMC=${3:-1}
R=$PWD; export PYTHONPATH=$R
OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_${CFG}_kt; rm -rf $D
timeout 400 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python $R/bench.py --config $CFG --steps 500 --warmup 100 --cpu-seconds 0 --no-others > $D.log 2>&1
grep '^{' $D.log | tail -1 > $OUT/${TAG}_${CFG}_bench_line.json
python $R/tools/summarize_profile.py $D $OUT/${TAG}_${CFG}_kernel_trace.md "${TAG}, config ${CFG}: rocprofv3 --kernel-trace --stats -- python bench.py --config ${CFG} --steps 500 --warmup 100 --cpu-seconds 0 --no-others"
[ "$PMC" = "1" ] || { cd $R; ls $OUT; exit 0; }
for C in FETCH_SIZE WRITE_SIZE; do
  D=$R/gpurun_out/pmc_${TAG}_${CFG}_$C; rm -rf $D
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d $D -o t --output-format csv -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --cpu-seconds 0 --no-roofline --no-others --settle-ms 0 > $D.log 2>&1
  python $R/tools/summarize_profile.py $D $OUT/${TAG}_${CFG}_pmc_$(echo $C | tr A-Z a-z).md "${TAG}, config ${CFG}: rocprofv3 --pmc $C --kernel-trace -- python bench.py --config ${CFG} --steps 10 --warmup 3 --no-others"
done
cd $R
python tools/make_pmc_traffic.py $CFG gpurun_out/pmc_${TAG}_${CFG}_FETCH_SIZE gpurun_out/pmc_${TAG}_${CFG}_WRITE_SIZE > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
ls $OUT
