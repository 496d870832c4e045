#!/bin/bash
# Round 3, config 5: whole-line finish (LDS-assembled stores) against the direct stores of round 2 (GP_HESS_DIRECT=1).
set -e
OUT=gpurun_out/r03_hess_lds
mkdir -p $OUT
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hess" > $OUT/pytest_hess.txt 2>&1 || { tail -30 $OUT/pytest_hess.txt; exit 1; }
tail -3 $OUT/pytest_hess.txt
for r in 0 1 2; do
  for prec in f64 f32; do
    for v in 0 1; do
      GP_HESS_DIRECT=$v python3 bench.py --workload c5 --precision $prec --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/b.json
      python3 -c "import json;d=json.load(open('$OUT/b.json'));print('round $r $prec direct=$v kernel_ms %.4f e_hess %.2e' % (d['roofline']['kernel_ms'], d['parity']['e_hess']))" | tee -a $OUT/ab.txt
    done
  done
done
GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_stamps.so python3 tools/hess_stamps.py > $OUT/stamps.txt 2>&1; cat $OUT/stamps.txt
for v in 0 1; do
  GP_HESS_DIRECT=$v rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$v -- python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2>$OUT/pmc_$v.err
  python3 - <<PY
import csv,glob
f=glob.glob('$OUT/pmc_write_$v/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'hessian' in r['Kernel_Name']]
vals=[float(r['Counter_Value']) for r in rows if r['Counter_Name']=='WRITE_SIZE']
print('direct=$v WRITE_SIZE per launch (KiB): n=%d mean=%.0f' % (len(vals), sum(vals)/len(vals)))
PY
done
