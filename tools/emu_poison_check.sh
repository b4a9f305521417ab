#!/bin/bash
# TEST INFRASTRUCTURE: the emulated-kernel tests with LDS filled with NaN bit patterns before every workgroup (on the GPU a
# workgroup finds whatever the previous one left in LDS, and 0 x NaN = NaN): a kernel that reads an LDS location it has not
# written - e.g. through a zero-weight MFMA operand - fails its parity test here.      bash tools/emu_poison_check.sh
cd "$(dirname "$0")/.."
HIPEMU_POISON_LDS=1 python -m pytest tests/test_emu_kernels.py -q "$@"
