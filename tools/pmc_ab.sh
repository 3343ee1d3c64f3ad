#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one bench config under environment variants (GPU box):  tools/pmc_ab.sh c3 "" "PDWT_SWTF_ALT=1"
CFG=$1; shift
R=$PWD; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    D=$R/gpurun_out/pmcab_${CFG}_${i}_$CNT; rm -rf $D
    env $V rocprofv3 --pmc $CNT --kernel-trace -d $D -o t --output-format csv -- python $R/bench.py --config $CFG --steps 6 --warmup 2 --cpu-seconds 0 --no-roofline --no-others --settle-ms 0 > $D.log 2>&1
    echo "== [$V] $CNT"
    python - $D $CNT <<'PY'
import csv, sys, glob, collections
d, cnt = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == cnt:
            acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:4]:
    mult = 2 if cnt == "FETCH_SIZE" else 1
    print("  %-60s n=%4d  avg %.1f MB per launch" % (k, len(v), mult * sum(v) / len(v) * 1024 / 1e6))
PY
  done
  i=$((i+1))
done
