#!/bin/bash
# SQ issue / stall counters and cache traffic of the scan prototype (tools/proto_scan_v5.hip), separate passes
mkdir -p gpurun_out/prof; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof/v5_sq1_$v -o pmc -- $R/build/proto_scan_v5 $v > $R/gpurun_out/prof/v5_sq1_$v.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/prof/v5_sq2_$v -o pmc -- $R/build/proto_scan_v5 $v > $R/gpurun_out/prof/v5_sq2_$v.log 2>&1
timeout 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof/v5_tc_$v -o pmc -- $R/build/proto_scan_v5 $v > $R/gpurun_out/prof/v5_tc_$v.log 2>&1
done
cd $R/gpurun_out/prof
python3 - <<'PY'
import csv, glob, collections, statistics
for d in sorted(glob.glob("v5_*_[01]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(d, k, c, f"median {statistics.median(v):.4g} n={len(v)}")
PY
