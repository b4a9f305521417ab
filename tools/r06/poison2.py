import sys, os, torch
sys.path.insert(0, os.getcwd())
from tests import test_gpu_blocks_conditioned as T
from tests import test_gpu_model as M
cases = [tuple(int(v) for v in c.split(",")) for c in os.environ.get("CASES", "48,48,32;96,96,16;96,48,32;192,192,8").split(";") if c]
for c in cases:
    T.test_unet_res_block_conditioned(*c)
try:
    M.test_segmamba_tiny_forward_golden(); print("cases", cases, "-> golden OK", flush=True)
except AssertionError as e:
    print("cases", cases, "-> golden FAILED", str(e)[:120], flush=True)
