#!/usr/bin/env python3
"""Is the packed-tile kernel (projected tables) bound by where its rows come from?  The C3 launch (524 288 Zipf pairs, KG adjacency) on
the real 106 389-entity tables against the same launch on a KG of the same degree statistics over 6 000 entities (3 x 1.5 MB of projected
tables + 1.5 MB of adjacency: everything stays in every 4 MB L2).  Development aid, GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops, synth
D, K, B = 64, 32, 524288
dev = torch.device("cuda:0")


def run(n_entity, n_item):
    d = dict(synth.DATASETS["last-fm_50core"], n_entity=n_entity, n_item=n_item)
    synth.DATASETS["probe"] = d
    case = synth.dataset_case("probe", K=K, B=B, seed=0, zipf=True, uniform_adj=False)
    g = torch.Generator(device=dev); g.manual_seed(0)
    nE, nR = case.n_entity, case.n_relation
    E = torch.rand((nE, D), device=dev, generator=g) - 0.5
    W = (torch.rand((3, D, D), device=dev, generator=g) - 0.5) / 8
    b = torch.rand((3, D), device=dev, generator=g) - 0.5
    t0, t1 = torch.rand(nR, device=dev, generator=g), torch.rand(nR, device=dev, generator=g)
    q = torch.rand((B, D), device=dev, generator=g) - 0.5
    ae = torch.from_numpy(case.adj_entity.astype("int32")).to(dev)
    ar = torch.from_numpy(case.adj_relation.astype("int32")).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    items = torch.from_numpy(case.items).to(dev)
    ws = ops.project_tables(E, W[0], W[1], b[0], b[1], W[2], b[2], K, True)
    fn = lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    it = items.long()
    ch = (enc_e[it] & 0xFFFFFF).long()
    real = ((enc_r[it] >> 16) & 0xFF) > 0
    rows = float((2 * cnt[it].double() + (cnt[ch].double() * real).sum(1)).mean())
    print(f"n_entity {nE:7d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us   distinct slots per row {float(cnt.float().mean()):.2f}   rows loaded per pair {rows:.1f}")


run(106389, 48091)
run(6000, 2712)
