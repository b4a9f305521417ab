#!/bin/bash
# timings and SQ issue counters of the forward convolution kernel variants (what bounds them: MFMA, VALU staging, LDS, waits)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cd $R; SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v "Warning\|amdgpu" | tee gpurun_out/r02_conv_fwd_variants.log; cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  PYTHONPATH=$R SEGM_TIME_CONV_ONLY=1 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof/conv_$tag -o pmc -- python $R/tools/gpu_chain_time.py > $R/gpurun_out/prof/conv_$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - $R <<'PY'
import csv, glob, sys, collections, statistics
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/prof/conv_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3d_k3_fwd" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:70], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    m = {c: statistics.median(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k} grid {g}\n   " + "  ".join(f"{c[3:]} {m[c] / wc:.3f}" if c not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU") else f"{c[3:]} {m[c]:.3g}" for c in sorted(m)))
PY
