#!/bin/bash
# issue / stall breakdown of the scan kernels (SQ block, one pass)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
bash -c 'true'
[ -f /tmp/pmc.py ] || sed -n '/^cat > \/tmp\/pmc.py/,/^PY$/p' tools/gpu_pmc_scan.sh | sed '1d;$d' > /tmp/pmc.py
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d gpurun_out/prof/pmc_SQ -o pmc -- python /tmp/pmc.py > gpurun_out/prof_pmc_SQ.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/prof_pmc_SQ.log
