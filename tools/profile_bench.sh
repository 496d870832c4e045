#!/bin/bash
# Profile one bench.py workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh <tag> <steps> [bench.py flags...]
# Writes gpurun_out/prof_<tag>/{bench.json, bench_under_rocprof.json, trace/, pmc_fetch/, pmc_write/, pmc_sq/}
# and the judged summaries gpurun_out/prof_<tag>/summary/*, which tools/rocprof_summary.py builds
# (copy those into profiles/).  PMC counters run in their own passes (kernel-trace/stats only).
set -e
TAG=$1; STEPS=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e "$@" > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-e2e "$@" > $OUT/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > /dev/null
python3 tools/rocprof_summary.py $OUT $TAG 10
