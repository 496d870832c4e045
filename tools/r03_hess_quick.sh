#!/bin/bash
# quick A/B of library variants on config 5 fp64 + stamps:  bash tools/r03_hess_quick.sh name=lib.so ...
set -e
OUT=gpurun_out/r03_hess_quick
mkdir -p $OUT
python3 tools/ab_bench.py --rounds 3 --args "--workload c5 --no-parity" "$@" > $OUT/ab.txt 2>&1
grep -v "^round" $OUT/ab.txt
if [ -f gp_emulator_amd/libgp_predict_hip_stamps.so ]; then
GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_stamps.so python3 tools/hess_stamps.py > $OUT/stamps.txt 2>&1; cat $OUT/stamps.txt
fi
