#!/bin/bash
# Whole-step FETCH / WRITE passes: bash tools/step_traffic.sh <tag> [pairs]
set -u
TAG=${1:-r03}
PAIRS=${2:-2}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/steptraffic_${TAG}_$c -o t -- python tools/step_traffic_target.py $PAIRS > gpurun_out/steptraffic_${TAG}_$c.log 2>&1
done
python tools/step_traffic.py gpurun_out/${TAG}_step_traffic.json gpurun_out/steptraffic_${TAG}_FETCH_SIZE gpurun_out/steptraffic_${TAG}_WRITE_SIZE $PAIRS
find gpurun_out/steptraffic_${TAG}_* -type f -size +1M -delete
