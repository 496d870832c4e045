#!/bin/bash
# SQ counters of the dominant kernel of a bench.py workload, two passes of 8 (run through gpurun):
#   bash tools/pmc_sq.sh <tag> [bench.py flags...]   ->  gpurun_out/pmc_<tag>/summary.txt
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e "$@" > /dev/null 2> $OUT/p$i.err
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
# dominant kernel = largest total duration
k = max((n for n in dur if 'gpk::' in n), key=lambda n: sum(dur[n]))
print('kernel', k[:110])
print('launches', len(dur[k]), 'mean ms under the counters %.4f' % (sum(dur[k]) / len(dur[k])))
for c, v in sorted(acc[k].items()):
    print('%-34s %.6g' % (c, sum(v) / len(v)))
PY
