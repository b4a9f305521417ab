#!/bin/bash
# Where do the waves of a whole training step WAIT?  One rocprofv3 --pmc pass over two eager steps: per kernel the share of its wave
# cycles spent parked (s_waitcnt / barrier), stalled at issue, and issuing - a detector for memory round trips on a critical path
# (the accumulate epilogues of round 4 showed up as `WAIT_ANY` 0.6 - 0.7).
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof/stepwait -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-configs --no-graph > $R/gpurun_out/prof/stepwait.log 2>&1
echo "rc=$?"
python3 - $R <<'PY' | tee $R/gpurun_out/r04_step_wait_pmc.log
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(R + "/gpurun_out/prof/stepwait/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:96]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
tot = sum(d["SQ_WAVE_CYCLES"] for d in agg.values()) or 1
print("# share of all wave cycles | parked (s_waitcnt, barrier) | issue-stalled | issuing | launches | kernel     (4 steps incl. warm-up, eager)")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:45]:
    wc = d["SQ_WAVE_CYCLES"] or 1
    print(f"{100 * wc / tot:5.1f} %  parked {d['SQ_WAIT_ANY'] / wc:.2f}  stalled {d['SQ_WAIT_INST_ANY'] / wc:.2f}  issuing {d['SQ_ACTIVE_INST_ANY'] / wc:.2f}  n={n[k]:4d}  {k}")
PY
