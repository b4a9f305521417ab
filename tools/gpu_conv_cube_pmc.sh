#!/bin/bash
# SQ counters of the cube kernels (forward and weight gradient at 768 -> 384 @16^3, 384 -> 192 @32^3, 768 -> 768 @8^3) (same passes as tools/history/conv_pmc.sh): separate --pmc passes, kernel trace only
# usage: bash tools/gpu_conv_cube_pmc.sh <tag>
TAG=${1:-r06_conv_cube_pmc}
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof/${TAG}_$i -o pmc -- python $R/tools/gpu_conv_cube_run.py > $R/gpurun_out/prof/${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $R $TAG <<'PY' | tee $R/gpurun_out/$TAG.log
import csv, glob, sys, collections, statistics
R, TAG = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/prof/" + TAG + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3d_k3_cube" in k and "reduce" not in k:
            agg[(k[:96], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
raw = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
       "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum")
for (k, g), d in sorted(agg.items()):
    m = {c: statistics.median(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    mf = m.get("SQ_INSTS_MFMA", 0) or 1
    print(f"{k} grid {g}")
    print("   per wave cycle: " + "  ".join(f"{c[3:]} {m[c] / wc:.3f}" for c in sorted(m) if c not in raw))
    print("   counts: " + "  ".join(f"{c} {m[c]:.4g}" for c in raw if c in m))
    if "SQ_INSTS_VALU" in m:
        print(f"   per MFMA: other VALU {(m['SQ_INSTS_VALU'] - mf) / mf:.2f}  SALU {m.get('SQ_INSTS_SALU', 0) / mf:.2f}  LDS {m.get('SQ_INSTS_LDS', 0) / mf:.2f}  VMEM {(m.get('SQ_INSTS_VMEM_RD', 0) + m.get('SQ_INSTS_VMEM_WR', 0)) / mf:.2f}")
    if "TCC_HIT_sum" in m:
        print(f"   L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
