#!/bin/bash
# A/B of an environment switch on config 5:  bash tools/r03_hess_env_ab.sh VAR "v0 v1" [precisions]
set -e
VAR=$1; VALS=$2; PRECS=${3:-"f64 f32"}
OUT=gpurun_out/r03_hess_env_ab
mkdir -p $OUT; rm -f $OUT/ab.txt
for r in 0 1 2; do
  for prec in $PRECS; do
    for v in $VALS; do
      env $VAR=$v python3 bench.py --workload c5 --precision $prec --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/b.json
      python3 -c "import json;d=json.load(open('$OUT/b.json'));print('round $r $prec $VAR=$v kernel_ms %.4f e_hess %.2e' % (d['roofline']['kernel_ms'], d['parity']['e_hess']))" | tee -a $OUT/ab.txt
    done
  done
done
