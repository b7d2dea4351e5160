#!/bin/bash
# Kernel timeline of ONE config-2 prove (headline driver): every kernel's start offset, duration and the idle gap in front of it, from rocprofv3 --kernel-trace.
# Output: gpurun_out/timeline.txt
R=$(pwd); O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cat > /tmp/tl_prove.py <<'PY'
import sys, os, time
R = os.environ["REPO"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from spartan2_amd import frontend, hip, host
inst = frontend.sha256_circuit(bytes(2048))
ctx = hip.Context(0); sn = host.SpartanSNARK(ctx, inst)
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
sn.prep_prove(tape)
for i in range(5): sn.prove(tape)
time.sleep(0.05)   # a visible gap in the trace in front of the prove that is listed
w, u, ph = sn.prove(tape)
print(ph)
PY
cd /tmp && export TMPDIR=/tmp
REPO=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python /tmp/tl_prove.py > $O/log.txt 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/tl/**/run_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the listed prove = the kernels between the last two gaps > 30 ms
starts = [int(r['Start_Timestamp']) for r in rows]
ends = [int(r['End_Timestamp']) for r in rows]
cuts = [i for i in range(1, len(rows)) if starts[i] - ends[i - 1] > 30_000_000]
lo, hi = (cuts[-1], len(rows)) if cuts else (0, len(rows))  # the prove behind the last pause
t0 = starts[lo]
out = open('gpurun_out/timeline.txt', 'w')
out.write("offset_us  dur_us  gap_us  kernel\n")
prev_end = t0
for i in range(lo, hi):
    n = rows[i]['Kernel_Name'].split('(')[0][:60]
    out.write(f"{(starts[i]-t0)/1e3:9.1f} {(ends[i]-starts[i])/1e3:7.1f} {(starts[i]-prev_end)/1e3:7.1f}  q{rows[i].get('Queue_Id','?')}  {n}\n")
    prev_end = max(prev_end, ends[i])
busy = sum(ends[i] - starts[i] for i in range(lo, hi))
out.write(f"kernels {hi-lo}, span {(max(ends[lo:hi])-t0)/1e3:.1f} us, sum of kernel durations {busy/1e3:.1f} us\n")
out.close()
print(open('gpurun_out/timeline.txt').read())
PY
rm -rf $O
