#!/bin/bash
# Round 3: the host-pointer pipeline with one stream per copy direction -- slab size x slots (run through gpurun)
for rounds in 1 2; do for slots in 3 4 6; do
  echo "GP_HOST_SLAB_ROUNDS=$rounds GP_HOST_SLOTS=$slots"
  GP_HOST_TRACE=1 GP_HOST_SLAB_ROUNDS=$rounds GP_HOST_SLOTS=$slots python3 tools/host_path_timing.py --quick --bind 2>&1 | tail -2
done; done
