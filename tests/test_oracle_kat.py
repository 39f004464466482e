"""CPU: analytic known-answer tests of the oracle (no second implementation needed).
SURVEY.md section 8(c)(3): properties that follow from the reference's equations."""
import copy

import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from oracle import equations_fp64, mirror_fp32


def base(**kw):
    d = dict(dim=8, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=5)
    d.update(kw)
    args = make_args(**d)
    case = synth.small_case(args, seed=21, zero_rows=4)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=22, random_agg_bias=True)
    return args, case, params


def fwd64(args, case, params):
    return equations_fp64.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                                  case.memories_h, case.memories_r, case.memories_t)


def fwd32(args, case, params):
    return mirror_fp32.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                               case.memories_h, case.memories_r, case.memories_t)


def test_aggregator_k1_is_identity_mix():
    # K=1: softmax over one neighbor is 1 and the mean over one element is the element
    D, B, N = 8, 3, 5
    g = torch.Generator().manual_seed(0)
    selfv, neigh = torch.randn(B, N, D, generator=g), torch.randn(B, N, 1, D, generator=g)
    rel, user = torch.randn(B, N, 1, D, generator=g), torch.randn(B, D, generator=g)
    agg, p = mirror_fp32.mix_neighbor_vectors_urh(selfv, user, neigh, rel, torch.randn(3 * D, 1, generator=g), B, D)
    assert torch.equal(p, torch.ones(B, N, 1))
    assert torch.allclose(agg, neigh[:, :, 0])


def test_zero_urh_weights_gives_uniform_attention():
    D, B, N, K = 8, 2, 3, 4
    g = torch.Generator().manual_seed(1)
    selfv, neigh = torch.randn(B, N, D, generator=g), torch.randn(B, N, K, D, generator=g)
    rel, user = torch.randn(B, N, K, D, generator=g), torch.randn(B, D, generator=g)
    agg, p = mirror_fp32.mix_neighbor_vectors_urh(selfv, user, neigh, rel, torch.zeros(3 * D, 1), B, D)
    assert torch.allclose(p, torch.full((B, N, K), 1.0 / K))
    # double normalisation: softmax weights AND reduce_mean (aggregators.py:144)
    assert torch.allclose(agg, neigh.mean(dim=2) / K, atol=1e-7)
    assert torch.allclose(mirror_fp32.mix_neighbor_vectors_no_ur(neigh), neigh.mean(dim=2))


def test_user_and_self_slices_of_urh_weights_cancel():
    # softmax shift invariance: the user and self terms of the logit are constant over k
    args, case, params = base()
    ref = fwd64(args, case, params)
    p2 = copy.deepcopy(params)
    D = args.dim
    rng = np.random.default_rng(5)
    for k in p2:
        if k.endswith("urh_weights"):
            p2[k][:D] += rng.normal(size=(D, 1)).astype(np.float32)
            p2[k][2 * D:] += rng.normal(size=(D, 1)).astype(np.float32)
    got = fwd64(args, case, p2)
    np.testing.assert_allclose(got.scores, ref.scores, rtol=1e-10, atol=1e-12)
    for a, b in zip(got.importance_list, ref.importance_list):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)


def test_identical_relations_give_uniform_attention():
    args, case, params = base()
    case.adj_relation[:] = 2
    out = fwd64(args, case, params)
    for p in out.importance_list:
        np.testing.assert_allclose(p, 1.0 / args.neighbor_sample_size, rtol=1e-12)


def test_identity_projection_with_zero_query():
    # W_e = I, b_e = 0, q = 0  =>  User_orient projection is the identity (model.py:277-279)
    args, case, params = base(ablation="no_kg_eh_uo")  # q = user embedding
    D = args.dim
    params["user_emb_matrix"][:] = 0
    for e in range(args.h_hop * args.n_mix_hop + 1):
        params[f"transfer_matrix_{e}"][:] = np.eye(D, dtype=np.float32)
        params[f"transfer_bias_{e}"][:] = 0
    a = fwd64(args, case, params)
    args2 = make_args(**dict(vars(args), ablation="no_uo_and_no_kg_eh_uo"))
    b = fwd64(args2, case, params)
    np.testing.assert_allclose(a.item_embeddings, b.item_embeddings, rtol=1e-12, atol=1e-14)


def test_zero_adjacency_row_means_entity0_relation0():
    args, case, params = base()
    z = np.nonzero((case.adj_entity == 0).all(axis=1))[0]
    assert len(z) >= 1
    case.items[0] = z[0]
    out = fwd32(args, case, params)
    assert (out.entities[1][0] == 0).all() and (out.relations[0][0] == 0).all()
    np.testing.assert_allclose(out.importance_list[0][0].numpy(), 1.0 / args.neighbor_sample_size, rtol=1e-6)


def test_permuting_children_leaves_score_unchanged():
    args, case, params = base()
    ref = fwd64(args, case, params)
    rng = np.random.default_rng(9)
    for x in range(case.n_entity):
        perm = rng.permutation(args.neighbor_sample_size)
        case.adj_entity[x] = case.adj_entity[x][perm]
        case.adj_relation[x] = case.adj_relation[x][perm]
    got = fwd64(args, case, params)
    np.testing.assert_allclose(got.scores, ref.scores, rtol=1e-10, atol=1e-12)


def test_key_addressing_matches_direct_formula():
    # (R h).v == h.(v R): the reassociation the HIP path uses (model.py:214-220)
    args, case, params = base()
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    b = 1
    E, RK = p["entity_emb_matrix"], p["relation_emb_KGE_matrix"]
    v = E[case.items[b]]
    for hop in range(args.p_hop):
        h = E[case.memories_h[hop][b]]
        r = case.memories_r[hop][b]
        direct = np.array([(RK[r[m]] @ h[m]) @ v for m in range(len(r))])
        V = np.einsum("i,rij->rj", v, RK)
        np.testing.assert_allclose(np.einsum("md,md->m", h, V[r]), direct, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("abl", ["all", "no_uor"])
def test_sum_then_project_equals_project_then_sum(abl):
    # the HIP deepest hop projects AFTER the weighted sum; exact by linearity since sum_k p_k = 1
    # (or = K without attention)
    args, case, params = base(ablation=abl)
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    D, K = args.dim, args.neighbor_sample_size
    rng = np.random.default_rng(3)
    rows, q = rng.normal(size=(K, D)), rng.normal(size=D)
    w = rng.random(K)
    w = w / w.sum() if abl == "all" else np.ones(K)
    W, b = p["transfer_matrix_2"], p["transfer_bias_2"]
    lhs = (w[:, None] * ((rows + q) @ W + b)).sum(0) / K
    rhs = ((w[:, None] * rows).sum(0) @ W + w.sum() * (q @ W + b)) / K
    np.testing.assert_allclose(lhs, rhs, rtol=1e-12, atol=1e-14)
