#!/bin/bash
# Round 3: the training-objective bench lines and kernel stats alone (the part of tools/profile_all.sh that changed).
OUT=gpurun_out/r03_train_profile; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "likelihood or train or learn" > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
python3 bench.py --workload train --steps 10 --warmup 3 > $OUT/train_f64_bench.json || exit 1
python3 bench.py --workload train --emulators 1 --steps 30 --warmup 5 > $OUT/train1_f64_bench.json || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 bench.py --workload train --steps 12 --warmup 4 > /dev/null 2>> $OUT/rocprof.log || exit 1
cp $(ls $OUT/trace_train/*/*kernel_stats.csv | head -1) $OUT/train_f64_kernel_stats.csv; rm -rf $OUT/trace_train
cat $OUT/train_f64_bench.json | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2101 sets', j['ms_per_step'], 'ms', j['value'])"
cat $OUT/train1_f64_bench.json | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 set', j['ms_per_step'], 'ms')"
head -5 $OUT/train_f64_kernel_stats.csv
