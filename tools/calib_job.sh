#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calib.hip) + the same two counters on the poly_ABC walk run alone: two PMC passes each, counters never
# combined with trace domains. Usage: tools/calib_job.sh r06  ->  gpurun_out/<round>_pmc_calibration.json
RD=${1:-r06}
R=$(pwd); O=$R/gpurun_out/calib_$RD; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $R/tools/fetch_calib > $O/cases.txt 2> $O/fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $R/tools/fetch_calib > /dev/null 2> $O/write.log
E2="python $R/tools/kernel_evidence.py"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pfetch -o run -- $E2 solo > $O/pfetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pwrite -o run -- $E2 solo > $O/pwrite.log 2>&1
cd $R
f() { find $O/$1 -name "run_counter_collection.csv" | head -1; }
python tools/calib_report.py --cases $O/cases.txt --fetch $(f fetch) --write $(f write) --polyabc-fetch $(f pfetch) --polyabc-write $(f pwrite) --out gpurun_out/${RD}_pmc_calibration.json
rm -rf $O/fetch $O/write $O/pfetch $O/pwrite
