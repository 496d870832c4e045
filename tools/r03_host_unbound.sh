#!/bin/bash
# the host-pointer path from a calling thread that is NOT next to the device: unbound, and pinned to the far socket
for rep in 0 1; do for v in 1 0; do
  echo "unbound GP_PIPE_DIRSTREAMS=$v"; GP_PIPE_DIRSTREAMS=$v python3 tools/host_path_timing.py --quick 2>&1 | tail -1
done; done
FAR=$(cat /sys/devices/system/node/node0/cpulist)
for v in 1 0; do
  echo "far socket (cpus $FAR) GP_PIPE_DIRSTREAMS=$v"; GP_PIPE_DIRSTREAMS=$v taskset -c $FAR python3 tools/host_path_timing.py --quick 2>&1 | tail -1
done
for v in 1 0; do
  echo "near (--bind) GP_PIPE_DIRSTREAMS=$v"; GP_PIPE_DIRSTREAMS=$v python3 tools/host_path_timing.py --quick --bind 2>&1 | tail -1
done
