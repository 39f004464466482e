"""Reference training step for the MVIN path (TEST INFRASTRUCTURE): loss of
src/model/MVIN/model.py:378-412 on top of oracle/mirror_fp32.py, gradients by torch
autograd, and tf.train.AdamOptimizer's update rule (model.py:414).

Loss pieces pinned to the reference graph (tests/golden/ref, oracle/__init__.py); gradients are a restatement, checked against finite differences
in tests/test_train_oracle.py.

Loss (model.py:379-412), with B = batch size:
  base  = mean_b sigmoid_cross_entropy(labels, scores)                               :379-380
  l2    = sum_hop ( sum(h_emb^2) + sum(t_emb^2) + sum(r_emb^2) )   [gathered rows]   :383-386
          (tf.reduce_mean of a scalar reduce_sum is that scalar)
        + l2_loss(relation_emb_matrix)                                               :388
        + [p_hop > 0] ( l2_loss(user_mlp_matrix) + l2_loss(user_mlp_bias)            :404
                        + l2_loss(LAST transfer matrix) + l2_loss(its bias)          :405
                        + sum_{n <= h_hop} l2_loss(transfer_matrix[n]) + bias )      :407-408
        + l2_loss(h_emb_item_mlp_matrix) + l2_loss(h_emb_item_mlp_bias)              :410
  l2agg = l2_loss(user_emb_matrix)                                                   :392
        + [not PS_only] sum_aggregators ( l2_loss(weights) + l2_loss(urh_weights) )  :393-396
        + sum_n ( l2_loss(enti_transfer_matrix[n]) + l2_loss(enti_transfer_bias[n]) ):400-401
  loss  = base + l2_weight * l2 + l2_agg_weight * l2agg                              :412
with tf.nn.l2_loss(x) = sum(x^2) / 2.
"""
import numpy as np
import torch

from . import mirror_fp32


def _l2(x):
    return (x * x).sum() / 2


def loss_from_params(args, p, adj_entity, adj_relation, users, items, labels, mem_h, mem_r, mem_t):
    """p: dict of torch tensors (requires_grad as wanted).  Returns (loss, pieces dict, forward out)."""
    out = mirror_fp32.forward(args, p, adj_entity, adj_relation, users, items, mem_h, mem_r, mem_t)
    labels = torch.as_tensor(np.asarray(labels)).to(out.scores.dtype)
    base = torch.nn.functional.binary_cross_entropy_with_logits(out.scores, labels, reduction="mean")
    l2 = torch.zeros((), dtype=out.scores.dtype)
    for hop in range(args.p_hop):
        h = p["entity_emb_matrix"][torch.as_tensor(mem_h[hop]).long()]
        t = p["entity_emb_matrix"][torch.as_tensor(mem_t[hop]).long()]
        r = p["relation_emb_KGE_matrix"][torch.as_tensor(mem_r[hop]).long()]
        l2 = l2 + (h * h).sum() + (t * t).sum() + (r * r).sum()
    l2 = l2 + _l2(p["relation_emb_matrix"])
    L = args.n_mix_hop * args.h_hop
    if args.p_hop > 0:
        l2 = l2 + _l2(p["user_mlp_matrix"]) + _l2(p["user_mlp_bias"])
        l2 = l2 + _l2(p[f"transfer_matrix_{L}"]) + _l2(p[f"transfer_bias_{L}"])
        for n in range(args.h_hop + 1):
            l2 = l2 + _l2(p[f"transfer_matrix_{n}"]) + _l2(p[f"transfer_bias_{n}"])
    l2 = l2 + _l2(p["h_emb_item_mlp_matrix"]) + _l2(p["h_emb_item_mlp_bias"])
    l2agg = _l2(p["user_emb_matrix"])
    if not args.PS_only:
        for k in p:
            if k.startswith("agg_") and (k.endswith("_weights")):   # weights and urh_weights
                l2agg = l2agg + _l2(p[k])
    for n in range(args.n_mix_hop):
        l2agg = l2agg + _l2(p[f"enti_transfer_matrix_{n}"]) + _l2(p[f"enti_transfer_bias_{n}"])
    loss = base + args.l2_weight * l2 + args.l2_agg_weight * l2agg
    return loss, {"base": base, "l2": l2, "l2agg": l2agg}, out


def loss_and_grads(args, params, adj_entity, adj_relation, users, items, labels, mem_h, mem_r, mem_t,
                   dtype=torch.float32):
    """Returns (loss float, grads dict of numpy arrays for every parameter that receives a
    gradient; parameters without one -- urh_bias, unused tables -- are absent, as TF's
    minimize() skips variables whose gradient is None)."""
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
    if dtype != torch.float32:
        raise NotImplementedError("the mirror is an fp32 graph")
    loss, pieces, out = loss_from_params(args, p, adj_entity, adj_relation, users, items, labels, mem_h, mem_r, mem_t)
    loss.backward()
    grads = {k: v.grad.numpy().copy() for k, v in p.items() if v.grad is not None}
    return float(loss.detach()), grads, {k: float(v.detach()) for k, v in pieces.items()}, out


class AdamRef(object):
    """tf.train.AdamOptimizer(lr) with TF1 defaults beta1=0.9, beta2=0.999, epsilon=1e-8
    (model.py:414).  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; var -= lr_t * m / (sqrt(v) + eps).
    Embedding tables receive IndexedSlices in TF; its _apply_sparse decays m and v of ALL rows and
    then updates all rows, which equals this dense rule with zero gradient on untouched rows."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0
        self.m = {k: np.zeros_like(np.asarray(v, dtype=np.float32)) for k, v in params.items()}
        self.v = {k: np.zeros_like(np.asarray(v, dtype=np.float32)) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        lr_t = np.float32(self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t))
        for k, g in grads.items():
            g = g.astype(np.float32)
            self.m[k] = np.float32(self.b1) * self.m[k] + np.float32(1 - self.b1) * g
            self.v[k] = np.float32(self.b2) * self.v[k] + np.float32(1 - self.b2) * g * g
            params[k] = (params[k] - lr_t * self.m[k] / (np.sqrt(self.v[k]) + np.float32(self.eps))).astype(np.float32)
        return params
