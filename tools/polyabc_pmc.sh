#!/bin/bash
# Counter passes for poly_ABC alone (one counter group per pass; never combined with other trace domains).
R=$(pwd); O=$R/gpurun_out/pabc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum FETCH_SIZE GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  for merged in 0 1; do
    SPARTAN_POLYABC_MERGED=$merged timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/$tag.$merged -o run -- python $R/tools/polyabc_prof.py > $O/$tag.$merged.log 2>&1
    f=$(find $O/$tag.$merged -name "run_counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python3 - "$f" "$merged" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "polyabc" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("merged=" + sys.argv[2], k[:40], {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
    else
      echo "no csv for $tag merged=$merged"; tail -3 $O/$tag.$merged.log
    fi
  done
done
cd $R
python tools/polyabc_prof.py; SPARTAN_POLYABC_MERGED=1 python tools/polyabc_prof.py
rm -rf $O/*/ 2>/dev/null
