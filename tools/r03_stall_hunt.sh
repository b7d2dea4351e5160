#!/bin/bash
# Round 3: the launch-ahead mailbox stall under many hardware queues (tools/concurrency_stress.py prints the mailbox diagnostics).
# Usage: tools/r03_stall_hunt.sh <proofs-per-context> [label]; writes gpurun_out/r03_stall_hunt_<label>.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
per=${1:-400}
out=gpurun_out/r03_stall_hunt_${2:-run}.txt
: > $out
run() {
  echo "=== $*" >> $out
  env "$@" timeout 1200 python tools/concurrency_stress.py --contexts 8 --proofs $per >> $out 2>&1
  echo "rc=$?" >> $out
}
run GPU_MAX_HW_QUEUES=24
run GPU_MAX_HW_QUEUES=24 SPARTAN_AHEAD_BUDGET=1000
run GPU_MAX_HW_QUEUES=24 SPARTAN_AHEAD_BUDGET=1000 SPARTAN_TAIL_BUDGET=128
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=4 SPARTAN_AHEAD_BUDGET=1000
run GPU_MAX_HW_QUEUES=4 SPARTAN_AHEAD_BUDGET=1000 SPARTAN_MAIL_MIRROR=0
cat $out
