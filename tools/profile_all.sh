#!/bin/bash
# Round-end measurement pass on the GPU box (through gpurun, from the repo root):
#   bash tools/profile_all.sh <round tag, e.g. r02>
# bench lines (un-profiled), rocprofv3 kernel stats per BASELINE config, PMC traffic for c2,
# the host-path timings, the strong-scaling rehearsal and the reference benchmark flow.
# Everything lands under gpurun_out/<tag>_final/; copy what is to be judged into profiles/.
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/${TAG}_final
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench lines" >&2
python3 bench.py > $OUT/c2_f64_bench.json
python3 bench.py --precision f32 --no-cpu-baseline > $OUT/c2_f32_bench.json
python3 bench.py --workload c3 --steps 5 --warmup 2 > $OUT/c3_f64_bench.json
python3 bench.py --workload c4 --steps 10 --warmup 3 > $OUT/c4_f64_bench.json
python3 bench.py --workload c5 --steps 20 --warmup 5 > $OUT/c5_f64_bench.json
python3 bench.py --workload c5 --precision f32 --steps 20 --warmup 5 > $OUT/c5_f32_bench.json
GP_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/c2_f64_2ranks_one_gpu.json
python3 bench.py --workload mv --steps 5 --warmup 2 > $OUT/mv_f64_bench.json
python3 bench.py --workload train --steps 10 --warmup 3 > $OUT/train_f64_bench.json
python3 bench.py --workload train --emulators 1 --steps 30 --warmup 5 > $OUT/train1_f64_bench.json
echo "== c2 profile (trace + PMC passes)" >&2
bash tools/profile_bench.sh c2_f64 200 > $OUT/profile_c2.log 2>&1
cp gpurun_out/prof_c2_f64/summary/* $OUT/
cp gpurun_out/prof_c2_f64/bench_under_rocprof.json $OUT/c2_f64_bench_under_rocprof.json
echo "== c3 / c4 / c5 kernel stats" >&2
for w in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline > $OUT/${w}_f64_bench_under_rocprof.json 2>> $OUT/rocprof.log
  mkdir -p gpurun_out/prof_${w}_f64 && rm -rf gpurun_out/prof_${w}_f64/trace && mv $OUT/trace_$w gpurun_out/prof_${w}_f64/trace
  cp $OUT/${w}_f64_bench_under_rocprof.json gpurun_out/prof_${w}_f64/bench_under_rocprof.json
  python3 tools/rocprof_summary.py gpurun_out/prof_${w}_f64 ${w}_f64 4 > /dev/null
  cp gpurun_out/prof_${w}_f64/summary/${w}_f64_kernel_stats.csv $OUT/
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 bench.py --workload train --steps 12 --warmup 4 > /dev/null 2>> $OUT/rocprof.log
cp $(ls $OUT/trace_train/*/*kernel_stats.csv | head -1) $OUT/train_f64_kernel_stats.csv; rm -rf $OUT/trace_train
echo "== host path" >&2
python3 tools/host_path_timing.py --c4 --bind > $OUT/host_path_timing.txt 2>&1
echo "== strong scaling rehearsal (ranks share the one GPU of this box)" >&2
python3 bench.py --workload c4 --scaling strong --total-rows 10000000 --steps 3 --warmup 1 > $OUT/strong_1rank_1e7.json
for n in 2 4; do
  # (bench.py starts its own ranks; GP_BENCH_SHARE_GPU=1: the ranks share this box's one GPU -- the line then says n_gpus 1, ranks n)
  GP_BENCH_SHARE_GPU=1 python3 bench.py --gpus $n --workload c4 --scaling strong --total-rows 10000000 --steps 3 --warmup 1 > $OUT/strong_${n}ranks_1e7.json 2>> $OUT/strong.err
done
python3 bench.py --workload c4 --scaling strong --total-rows 100000000 --steps 2 --warmup 1 > $OUT/strong_1rank_1e8.json
echo "== reference benchmark flow" >&2
python3 examples/benchmark.py > $OUT/reference_benchmark_flow.txt 2>&1
ls $OUT >&2
