#!/bin/bash
# A/B of library builds on the materialised-output call, interleaved so that clock drift hits both:
#   tools/ab_matrix.sh "<bench_matrix args>" libA.so libB.so [...]     (paths relative to the repo root)
ARGS=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%s round %d: " "$lib" $round
    STORM_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_matrix.py --ops and --reps 20 $ARGS 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'
  done
done
