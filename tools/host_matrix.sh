# A/B runs of tools/host_path_timing.py over the host-path knobs (one GPU box, one process each)
mkdir -p gpurun_out/r02b
rm -f gpurun_out/r02b/matrix.log
for cfg in "0 3 0" "1 3 0" "1 4 0" "1 6 0" "1 4 2" "1 3 2"; do
  set -- $cfg
  echo "=== GP_HOST_STREAMS=$1 GP_HOST_SLOTS=$2 GP_HOST_SKIP=$3" >> gpurun_out/r02b/matrix.log
  GP_HOST_STREAMS=$1 GP_HOST_SLOTS=$2 GP_HOST_SKIP=$3 GP_HOST_TRACE=1 timeout -k 10 120 python tools/host_path_timing.py --quick >> gpurun_out/r02b/matrix.log 2>&1 || exit 1
done
