#!/bin/bash
# What shader clock do the kernels run at?  rocm-smi sampled every 0.2 s while bench.py loops the c2 kernel (fp64, then fp32)
# for ~6 s each; idle readings first.   bash tools/clock_under_load.sh > gpurun_out/clock_under_load.txt
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^$" | head -30
for prec in f64 f32; do
  echo "== bench.py --workload c2 --precision $prec --steps 5000 running =="
  python3 bench.py --workload c2 --precision $prec --steps 5000 --warmup 5 --no-cpu-baseline --no-e2e > /tmp/b_$prec.json 2>/dev/null &
  pid=$!
  sleep 4     # (library load + inputs)
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
    sleep 0.4
  done
  wait $pid
  python3 -c "import json; j=json.loads(open('/tmp/b_$prec.json').read().strip().splitlines()[-1]); print('kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
done
