#!/bin/bash
# rocprofv3 kernel traces of a C2-only prove loop (headline driver, then the reference-order driver) and of the same kernels run one at a time, plus
# the two PMC passes of the prove loop (separate runs, counters never combined with trace domains). Usage: tools/profile_job.sh r05 - summaries under
# gpurun_out/ as <round>_kernel_stats.md, <round>_pmc_traffic.json, <round>_kernel_sites.json, <round>_kernel_stats_rocprof.csv (copied into profiles/ by hand).
RD=${1:-r05}
R=$(pwd); O=$R/gpurun_out/prof_$RD; rm -rf $O; mkdir -p $O
E="python $R/tools/kernel_evidence.py"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prove -o run -- $E prove > $O/prove.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/reford -o run -- $E prove --reference-order > $O/reford.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -o run -- $E solo > $O/solo.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $E prove > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $E prove > $O/write.log 2>&1
cd $R
d() { dirname $(find $O/$1 -name "run_$2.csv" | head -1); }
python tools/kernel_report.py --round $RD --prove $(d prove kernel_trace) --reford $(d reford kernel_trace) --solo $(d solo kernel_trace) --fetch $(d fetch counter_collection) \
  --write $(d write counter_collection) --bench-json gpurun_out/${RD}_bench.json --out gpurun_out/${RD}_kernel_stats.md --pmc-json gpurun_out/${RD}_pmc_traffic.json --sites-json gpurun_out/${RD}_kernel_sites.json > gpurun_out/${RD}_profile.log 2>&1
cp $(d prove kernel_trace)/run_kernel_stats.csv gpurun_out/${RD}_kernel_stats_rocprof.csv 2>/dev/null
cp $(d reford kernel_trace)/run_kernel_stats.csv gpurun_out/${RD}_kernel_stats_rocprof_reference_order.csv 2>/dev/null
for f in $O/*.log; do echo "== $f"; tail -n 6 $f; done >> gpurun_out/${RD}_profile.log 2>&1
rm -rf $O
tail -n 40 gpurun_out/${RD}_profile.log
