#!/usr/bin/env python3
"""Development aid (GPU box): the kernel over static user records against the bucketing kernel, shape by shape, each in its own process."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from mvin_amd import ops, synth
    P, Nm, nR, n_user, B = map(int, sys.argv[1:6])
    D, n_entity = 64, 5000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    E = torch.rand((n_entity, D), device=dev, generator=g) - 0.5
    R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
    w = torch.rand(D, device=dev, generator=g) - 0.5
    uts = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=B)).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    a = torch.zeros((B, (P + 1) * D), device=dev); b = torch.zeros_like(a)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, a, (P + 1) * D, nR)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, b, (P + 1) * D, nR, records=rec)
    torch.cuda.synchronize()
    print("max diff hset %.3g reads %.3g" % ((a[:, :D] - b[:, :D]).abs().max().item(), (a[:, D:] - b[:, D:]).abs().max().item()))
else:
    for shape in ["2 64 9 5 40", "2 64 9 5 3000", "2 64 9 100 1500", "2 64 9 300 1300", "2 64 9 2000 9000"]:
        r = subprocess.run([sys.executable, __file__] + shape.split(), capture_output=True, text=True)
        print(shape, "->", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], (r.stderr.strip().splitlines() or [""])[-1][:150])
