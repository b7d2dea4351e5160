#!/bin/bash
# A/B of library builds on one box: tools/ab/<name>/*.so copied over spartan2_amd/lib, bench legs interleaved. Usage: tools/ab/run.sh <outdir> <steps> <reps> name...
out=gpurun_out/$1; steps=$2; reps=$3; shift 3
mkdir -p $out
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    cp tools/ab/$v/*.so spartan2_amd/lib/
    python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-sharded --concurrent 0 > $out/$v.$rep.json 2> $out/$v.$rep.err
  done
done
