#!/bin/bash
# A/B of two library builds on one box: tools/ab/<name>/*.so copied over spartan2_amd/lib, bench legs interleaved
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2 3 4; do
  for v in "$@"; do
    cp tools/ab/$v/*.so spartan2_amd/lib/
    python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-sharded --concurrent 0 > $out/$v.$rep.json 2> $out/$v.$rep.err
  done
done
