#!/bin/bash
# SQ issue / stall counters of the scan kernels for one library variant: tools/gpu_scan_pmc.sh <lib.so> <tag>   (env is passed through)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_$2 -o pmc -- python $R/tools/gpu_scan_ab.py $R/$1 > $R/gpurun_out/prof/pmc_$2.log 2>&1
python3 - $R/gpurun_out/prof/pmc_$2 <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "segm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: statistics.median(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"{k}\n    VALU-busy cycles per SIMD {m.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024:9.0f}   busy cycles {m.get('SQ_BUSY_CYCLES', 0):9.0f}  VALU instr {m.get('SQ_INSTS_VALU', 0):.3g}"
          f"   of wave-cycles: VALU {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  any {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  wait_inst {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  waitcnt {m.get('SQ_WAIT_ANY', 0) / wc:.3f}")
PY
