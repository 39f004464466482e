#!/usr/bin/env python3
"""us per mvin_group_pairs_by_user call over batch sizes (set MVIN_GROUP_SMALL=0/1 to force a form)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mvin_amd import ops
n_user = int(sys.argv[1]) if len(sys.argv) > 1 else 23553
for B in (512, 4096, 16384, 65536, 131072, 262144, 524288):
    users = torch.randint(0, n_user, (B,), device="cuda:0")
    for _ in range(5):
        ops.group_pairs_by_user(users, n_user=n_user)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.group_pairs_by_user(users, n_user=n_user)
    e1.record()
    torch.cuda.synchronize()
    print(f"n_user {n_user} B {B:7d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call", flush=True)
