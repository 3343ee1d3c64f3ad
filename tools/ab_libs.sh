export PYTHONPATH=$PWD
cp -r pdwt_amd/lib lib_new
for rep in 1 2 3; do
  for v in old new; do
    rm -rf pdwt_amd/lib; cp -r lib_$v pdwt_amd/lib
    echo -n "$v: "; timeout 200 python bench.py --config c4 --steps 200 --warmup 20 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
