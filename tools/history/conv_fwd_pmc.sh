#!/bin/bash
# Round 5: L2 hit rate / HBM fetch of the chained forward kernels (separate --pmc passes)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  PYTHONPATH=$R SEGM_TIME_CONV_ONLY=1 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof/f5_$i -o pmc -- python $R/tools/gpu_chain_time.py > $R/gpurun_out/prof/f5_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $R <<'PY' | tee $R/gpurun_out/r05_conv_fwd_pmc.log
import csv, glob, sys, collections, statistics
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/prof/f5_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3d_k3_fwd48" in k:
            agg[(k[:76], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    m = {c: statistics.median(v) for c, v in d.items()}
    print(f"{k} grid {g}")
    print("   " + "  ".join(f"{c} {m[c]:.4g}" for c in sorted(m)))
    if "TCC_HIT_sum" in m:
        print(f"   L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
