#!/bin/bash
# Round 5, call 2: ablations of the weight-gradient row loop + L2 / HBM counters for v1 and r5 (separate --pmc passes)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT
timeout 600 python tools/history/r05/wgrad_abl.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tee gpurun_out/r05_wgrad_abl3.log
cd /tmp; export TMPDIR=/tmp
i=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  for v in "ct ipw1" "pipe ipw1"; do
    WG_ONLY="$v" PYTHONPATH=$R timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof/w7_${v// /_}_$i -o pmc -- python $R/tools/history/r05/wgrad_abl.py > $R/gpurun_out/prof/w7_${v// /_}_$i.log 2>&1
    echo "pass $i $v rc=$?"
  done
done
python3 - $R <<'PY' | tee $R/gpurun_out/r05_wgrad_pmc3.log
import csv, glob, sys, collections, statistics
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/prof/w7_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3d_k3_wgrad" in k and "reduce" not in k:
            agg[(k[:70], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    m = {c: statistics.median(v) for c, v in d.items()}
    print(f"{k} grid {g}")
    print("   " + "  ".join(f"{c} {m[c]:.4g}" for c in sorted(m)))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        print("   per wave cycle: " + "  ".join(f"{c[3:]} {m[c] / wc:.3f}" for c in sorted(m) if c.startswith("SQ_") and c != "SQ_WAVE_CYCLES"))
    if "TCC_HIT_sum" in m:
        print(f"   L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
