#!/bin/bash
# End-of-round refresh on the GPU box (through gpurun, from the repo root):  tools/refresh_profiles.sh <tag>
#   1. the driver's command, unprofiled                     -> gpurun_out/refresh_<tag>/bench_unprofiled.json
#   2. tools/profile_default.sh (trace + counter passes)    -> gpurun_out/prof_<tag>/
# Afterwards, locally: tools/pmc_traffic.py ... --merge, tools/pmc_summary.py, copy into profiles/.
set -e
TAG=$1
mkdir -p gpurun_out/refresh_$TAG
python3 bench.py > gpurun_out/refresh_$TAG/bench_unprofiled.json 2> gpurun_out/refresh_$TAG/bench_unprofiled.err
cat gpurun_out/refresh_$TAG/bench_unprofiled.json
bash tools/profile_default.sh $TAG > gpurun_out/refresh_$TAG/profile.log 2>&1
tail -3 gpurun_out/refresh_$TAG/profile.log
