#!/bin/bash
# Round 3, config 5 fp64: timing-only ablations of hessian_win_kernel + in-kernel stamps (run through gpurun).
set -e
OUT=gpurun_out/r03_hess_ablate
mkdir -p $OUT
L=gp_emulator_amd/libgp_predict_hip
python3 tools/ab_bench.py --rounds 3 --args "--workload c5 --no-parity" base=$L.so nobar=${L}_nobar.so nostore=${L}_nostore.so nofinish=${L}_nofinish.so nodma=${L}_nodma.so nobarnodma=${L}_nobarnodma.so > $OUT/ab.txt 2>&1
GP_PREDICT_LIB=${L}_stamps.so python3 tools/hess_stamps.py > $OUT/stamps.txt 2>&1
tail -8 $OUT/ab.txt; cat $OUT/stamps.txt
