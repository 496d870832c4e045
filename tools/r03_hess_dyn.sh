#!/bin/bash
set -e
OUT=gpurun_out/r03_hess_dyn
mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hess" > $OUT/pytest_hess.txt 2>&1 || { tail -30 $OUT/pytest_hess.txt; exit 1; }
tail -2 $OUT/pytest_hess.txt
for r in 0 1 2; do
  for prec in f64 f32; do
    for v in 0 1; do
      GP_HESS_STATIC=$v python3 bench.py --workload c5 --precision $prec --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/b.json
      python3 -c "import json;d=json.load(open('$OUT/b.json'));print('round $r $prec static=$v kernel_ms %.4f e_hess %.2e' % (d['roofline']['kernel_ms'], d['parity']['e_hess']))" | tee -a $OUT/ab.txt
    done
  done
done
GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_stamps.so python3 tools/hess_stamps.py > $OUT/stamps.txt 2>&1; cat $OUT/stamps.txt
GP_HESS_STATIC=1 GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_stamps.so python3 tools/hess_stamps.py > $OUT/stamps_static.txt 2>&1; tail -1 $OUT/stamps_static.txt
