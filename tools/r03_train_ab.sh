#!/bin/bash
# Round 3: training kernel variants side by side on one box, interleaved (bench.py --workload train, wall ms per call).
#   usage: tools/r03_train_ab.sh name=lib.so [name=lib.so ...]      (libs built with gp_emulator_amd.build --define ... --lib ... --only train)
out=gpurun_out/r03_train_ab; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "likelihood or train or learn" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for r in 0 1 2; do
  for v in "$@"; do
    name=${v%%=*}; lib=${v#*=}
    for e in 2101 1; do
      GP_PREDICT_LIB=$lib timeout -k 10 120 python bench.py --workload train --emulators $e --steps 20 --warmup 3 > $out/b.json 2>$out/err.txt || { cat $out/err.txt; exit 1; }
      python - "$name" "$e" "$r" <<'PY' | tee -a $out/ab.txt
import json,sys
j=json.loads(open("gpurun_out/r03_train_ab/b.json").read().strip().splitlines()[-1])
print("round %s %-10s sets=%-5s ms_per_step %.4f" % (sys.argv[3], sys.argv[1], sys.argv[2], j["ms_per_step"]))
PY
    done
  done
done
