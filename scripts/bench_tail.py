#!/usr/bin/env python3
"""us per mvin_l2_tail_fwd call at dim 64 on 524 288 pairs (MVIN_TAIL_FLASH=0: the tile-image kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_tail import _inputs
from mvin_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
args = list(_inputs(B, 64, 106389, seed=1))
args[3] = args[2]            # q IS user_o in the default wiring
for _ in range(3):
    ops.l2_tail(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.l2_tail(*args)
e1.record(); torch.cuda.synchronize()
print(f"l2_tail B={B} MVIN_TAIL_FLASH={os.environ.get('MVIN_TAIL_FLASH', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
