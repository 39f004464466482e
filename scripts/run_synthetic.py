#!/usr/bin/env python3
"""End-to-end run on synthetic data, everything on the GPU: KG triples + interactions ->
CSR -> sampled adjacency + ripple sets -> training epochs (MVIN.train semantics) -> CTR and top-K
evaluation.  The counterpart of `bash src/bash/mvin_*.sh` for the part of the reference this
repository rebuilds (data files of the reference are not shipped, so a learnable synthetic
signal stands in: label = 1 iff the item shares a KG neighbor with one of the user's items)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import data_prep, harness, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN


def synthetic_interactions(kg, n_user, n_item, n_entity, per_user, rng):
    """Users like items of a few 'taste' entities: an item is positive for a user iff it is a KG
    neighbor of one of the user's taste entities."""
    nbr = {}
    for h, r, t in kg.tolist():
        if h < n_item:
            nbr.setdefault(t, set()).add(h)
        if t < n_item:
            nbr.setdefault(h, set()).add(t)
    hubs = [e for e, s in nbr.items() if len(s) >= 3]
    rows = []
    for u in range(n_user):
        taste = rng.choice(hubs, size=2, replace=False)
        pos = set().union(*[nbr[e] for e in taste])
        pos = rng.permutation(list(pos))[:per_user]
        neg = rng.choice(n_item, size=len(pos))
        neg = [i for i in neg if i not in set(pos)]
        rows += [(u, i, 1) for i in pos] + [(u, i, 0) for i in neg]
    d = np.array(rows, dtype=np.int64)
    rng.shuffle(d)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-entity", type=int, default=4000); ap.add_argument("--n-item", type=int, default=600)
    ap.add_argument("--n-user", type=int, default=300); ap.add_argument("--n-relation", type=int, default=6)
    ap.add_argument("--dim", type=int, default=16); ap.add_argument("--fanout", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=4); ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    args = make_args(dataset="synthetic", dim=a.dim, neighbor_sample_size=a.fanout, h_hop=2, n_mix_hop=1, p_hop=2,
                     n_memory=16, batch_size=a.batch, lr=5e-3, l2_weight=1e-6, l2_agg_weight=1e-6)
    kg = synth.synth_kg(a.n_entity, a.n_relation, 10.0, seed=a.seed + 1)
    data = synthetic_interactions(kg, a.n_user, a.n_item, a.n_entity, 24, rng)
    n = len(data)
    train, evald, test = data[: int(0.6 * n)], data[int(0.6 * n): int(0.8 * n)], data[int(0.8 * n):]
    t0 = time.time()
    csr = data_prep.build_csr(kg, a.n_entity)                                  # construct_kg
    adj_e, adj_r = data_prep.construct_adj(csr, a.n_entity, a.fanout, seed=1)  # contruct_random_adj
    uts = data_prep.get_user_triplet_set(csr, data_prep.history_csr(train, a.n_user), a.n_user, args.p_hop,
                                         args.n_memory, seed=2)                # get_user_triplet_set
    torch.cuda.synchronize()
    print(f"inputs built on the GPU in {time.time() - t0:.2f}s: adjacency {tuple(adj_e.shape)}, ripple sets {tuple(uts.shape)}")
    model = MVIN(args, a.n_user, a.n_entity, a.n_relation, adj_e, adj_r, device="cuda:0", seed=a.seed)
    feeder = harness.DeviceFeeder(model, uts.cpu().numpy())
    out = {}
    for ep in range(a.epochs):
        losses = harness.train_epoch_device(feeder, train, a.batch, rng=rng)
        _, _, _, auc, acc, f1 = harness.ctr_eval_device(feeder, evald, a.batch)
        print(f"epoch {ep}: loss {np.mean(losses):.4f}  eval auc {auc:.4f} acc {acc:.4f} f1 {f1:.4f}")
        out.setdefault("loss", []).append(float(np.mean(losses)))
        out.setdefault("auc", []).append(auc)
    users, tr, ev, te, item_set, k_list = harness.topk_settings(train, evald, test, a.n_item, user_num=50)
    prec, rec, ndcg, _, _ = harness.topk_eval_device(feeder, users, tr, ev, te, item_set, k_list, a.batch, mode="test")
    print("top-K test: recall@10 %.4f  ndcg@10 %.4f  precision@10 %.4f" % (rec[3], ndcg[3], prec[3]))
    out["recall@10"] = rec[3]
    return out


if __name__ == "__main__":
    main()
