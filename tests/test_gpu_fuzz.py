"""-m gpu: seeded random sweep over the configuration space of the path (dim, fan-out, depth, mix blocks, ripple hops,
memories, relations, batch, ablation preset, table dtype, feed form) against the fp32 mirror and the fp64 equations.
The hand-picked shapes of the other parity tests pin every kernel instance; this one looks for interactions nobody
thought of (a ragged batch with a depth-3 tree, a preset without attention on the D = 16 kernel, one pair per user
in the grouped kernel, ...).  Sizes are kept where the CPU oracles finish in well under a second per case."""
import os

import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args
from mvin_amd.params import init_params

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("MVIN_FUZZ_CASES", "48"))      # a longer campaign: MVIN_FUZZ_CASES=1000 pytest tests/test_gpu_fuzz.py
OFFSET = int(os.environ.get("MVIN_FUZZ_OFFSET", "0"))               # first case number (fresh draws for a campaign)


def _draw(i):
    rng = np.random.default_rng(9000 + i)
    D = int(rng.choice([8, 12, 16, 16, 32, 32, 64, 64, 128]))
    K = int(rng.choice([2, 3, 4, 8, 8, 16, 32, 5]))
    H = int(rng.choice([1, 2, 2, 2, 3]))
    M = int(rng.choice([1, 1, 1, 2]))
    while K ** (H * M) > 4096:                      # rows per pair the oracles gather
        if M > 1:
            M = 1
        elif H > 1:
            H -= 1
        else:
            K = 4
    P = int(rng.choice([1, 2, 2, 3]))
    Nm = int(rng.choice([3, 4, 8, 16, 32, 64]))
    nR = int(rng.choice([2, 5, 9, 12, 39]))
    B = int(rng.choice([1, 2, 7, 16, 33, 64, 129]))
    n_user = int(rng.choice([1, 3, 17, 200]))
    abl = str(rng.choice(sorted(ABLATIONS)))
    tdt = "bf16" if (rng.random() < 0.25 and D % 8 == 0) else "f32"
    feed = str(rng.choice(["pairs", "users", "users_grouped"]))
    fused = bool(rng.random() < 0.8)
    # a second, independent stream for the switches added later (the draws above keep their values)
    rng2 = np.random.default_rng(50000 + i)
    hoist = [False, False, True, "step"][int(rng2.integers(0, 4))]       # entity-table mode
    ids32 = bool(rng2.random() < 0.3)                                   # int32 user / item ids instead of int64
    probs = bool(rng2.random() < 0.25)                                  # attention outputs requested (model.py:319-323)
    twice = bool(rng2.random() < 0.3)                                   # score twice on one model (cached state)
    return dict(D=D, K=K, H=H, M=M, P=P, Nm=Nm, nR=nR, B=B, n_user=n_user, abl=abl, tdt=tdt, feed=feed, fused=fused,
                hoist=hoist, ids32=ids32, probs=probs, twice=twice)


@pytest.mark.parametrize("i", range(OFFSET, OFFSET + N_CASES))
def test_random_configuration(i, hip_lib):
    from mvin_amd.model import MVIN
    c = _draw(i)
    args = make_args(dim=c["D"], neighbor_sample_size=c["K"], h_hop=c["H"], n_mix_hop=c["M"], p_hop=c["P"],
                     n_memory=c["Nm"], batch_size=c["B"], ablation=c["abl"])
    n_entity = 150 + 37 * (i % 5)
    case = synth.small_case(args, n_user=c["n_user"], n_entity=n_entity, n_relation=c["nR"], seed=9100 + i, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=9200 + i, random_agg_bias=True)
    oracle_params = params
    if c["tdt"] == "bf16":
        oracle_params = dict(params, entity_emb_matrix=torch.from_numpy(params["entity_emb_matrix"]).to(torch.bfloat16).float().numpy())
    # the users feeds read user_triplet_set[user]: make the per-pair arrays of the oracle exactly those rows
    uts = synth.ripple_sets(case.n_user, case.n_entity, case.n_relation, max(1, c["P"]), c["Nm"], seed=9300 + i)
    if c["feed"] != "pairs":
        case.memories_h, case.memories_r, case.memories_t = synth.memories_for(uts, case.users)
    m, e = run_oracles(args, case, oracle_params)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0", fused=c["fused"], table_dtype=c["tdt"], hoist=c["hoist"])
    dev = model.device
    u_d, i_d = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    if c["ids32"]:
        u_d, i_d = u_d.to(torch.int32), i_d.to(torch.int32)
    mem = [[torch.from_numpy(x).to(dev) for x in lst] for lst in (case.memories_h, case.memories_r, case.memories_t)]
    uts_d = torch.from_numpy(uts).to(dev)

    def score():
        if c["feed"] == "pairs":
            return model.forward_device(u_d, i_d, *mem, want_probs=c["probs"])
        model.group_min_pairs_per_user = 0 if c["feed"] == "users_grouped" else 10 ** 9
        return model.forward_users(u_d, i_d, uts_d, want_probs=c["probs"])
    out = score()
    if c["twice"]:
        first = out.scores.clone()
        out = score()
        assert torch.equal(first, out.scores), f"second call differs from the first, case {i} {c}"
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-6) if c["tdt"] == "f32" else dict(rtol=1e-5, atol=2e-6)
    what = f"case {i} {c}"
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), f"scores vs fp32 mirror, {what}", **tol)
    assert_close(out.user_o.cpu().numpy(), m.user_o.numpy(), f"user_o, {what}", **tol)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), f"item_embeddings, {what}", **tol)
    if c["probs"]:
        # importance_list of the last mix block's first aggregator pass (model.py:294,304,319-323), per hop
        assert len(out.importance_list) == len(m.importance_list), what
        for hop, (pg, pm) in enumerate(zip(out.importance_list, m.importance_list)):
            if pm is None:
                assert pg is None, f"importance_list[{hop}] given but the mirror has none, {what}"
            else:
                assert_close(pg.cpu().numpy(), pm.numpy(), f"importance_list[{hop}], {what}", rtol=1e-5, atol=1e-6)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 2e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}, {what}"


N_GRAPH = int(os.environ.get("MVIN_GRAPH_FUZZ_CASES", "8"))


@pytest.mark.parametrize("i", range(OFFSET, OFFSET + N_GRAPH))
def test_random_configuration_hipgraph_replay(i, hip_lib):
    """graph.GraphedScorer on a random configuration: the pass captured once, replayed on two NEW batches, must give
    the scores of eager launches on the same inputs bit for bit."""
    from mvin_amd.graph import GraphedScorer
    from mvin_amd.model import MVIN
    c = _draw(30000 + i)
    B = c["B"]
    args = make_args(dim=c["D"], neighbor_sample_size=c["K"], h_hop=c["H"], n_mix_hop=c["M"], p_hop=c["P"],
                     n_memory=c["Nm"], batch_size=B, ablation=c["abl"])
    big = make_args(**dict(vars(args), batch_size=2 * B))
    case = synth.small_case(big, n_user=c["n_user"], n_entity=150 + 37 * (i % 5), n_relation=c["nR"], seed=39100 + i, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=39200 + i, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0", fused=c["fused"], table_dtype=c["tdt"], hoist=c["hoist"])
    dev = model.device
    scorer = GraphedScorer(model, B)
    for s in range(2):
        sl = slice(s * B, (s + 1) * B)
        u = torch.from_numpy(np.ascontiguousarray(case.users[sl])).to(dev)
        it = torch.from_numpy(np.ascontiguousarray(case.items[sl])).to(dev)
        mem = [[torch.from_numpy(np.ascontiguousarray(x[sl])).to(dev) for x in lst]
               for lst in (case.memories_h, case.memories_r, case.memories_t)]
        got = scorer(u, it, *mem).scores.clone()
        ref = model.forward_device(u, it, *mem).scores
        torch.cuda.synchronize()
        assert torch.equal(got, ref), f"replay {s} differs from eager, case {i} {c}"


N_LARGE = int(os.environ.get("MVIN_LARGE_FUZZ_CASES", "6"))


@pytest.mark.parametrize("i", range(OFFSET, OFFSET + N_LARGE))
def test_random_large_shapes_fused_against_per_level(i, hip_lib):
    """Shapes and batches far beyond what the CPU oracles finish (ragged batches of thousands of pairs, tables of up to
    200 000 rows, fan-outs up to 64): the fused path (persistent grids, tile pipelines, grid-stride tails) against the
    per-level kernels -- an independent HIP implementation that the small cases pin to the oracle."""
    from mvin_amd.model import MVIN
    rng = np.random.default_rng(60000 + i)
    D = int(rng.choice([16, 32, 64, 128]))
    K = int(rng.choice([4, 8, 16, 32, 64]))
    H = 3 if (K <= 8 and rng.random() < 0.3) else 2
    rows = sum(K ** e for e in range(H + 1))
    bmax = max(1, int(1.5e9 // (K ** H * D * 4)))                     # the per-level path materialises [B, K^H, D]
    B = int(min(bmax, rng.choice([1, 3, 63, 64, 65, 257, 1000, 4099, 20000])))
    n_entity = int(rng.choice([300, 5000, 200000]))
    n_user = int(rng.choice([1, 50, 3000]))
    P = int(rng.choice([1, 2]))
    Nm = int(rng.choice([8, 16, 64]))
    nR = int(rng.choice([3, 9, 39]))
    tdt = "bf16" if rng.random() < 0.3 else "f32"
    feed = str(rng.choice(["pairs", "users", "users_grouped"]))
    abl = str(rng.choice(["all", "all", "no_uo", "no_uor", "no_ps_o_ft", "ho_only"]))
    what = f"case {i}: D={D} K={K} H={H} B={B} nE={n_entity} nU={n_user} P={P} Nm={Nm} nR={nR} {tdt} {feed} {abl} rows/pair={rows}"
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=abl)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, K, seed=61000 + i)
    repeats = bool(rng.random() < 0.6)        # rows as contruct_random_adj builds them for low-degree entities (:383-384)
    if repeats:
        pool_e, pool_r = rng.integers(0, n_entity, (n_entity, 8)), rng.integers(0, nR, (n_entity, 8))
        deg = rng.integers(1, 9, n_entity)
        pick = (rng.random((n_entity, K)) * deg[:, None]).astype(np.int64)
        low = rng.random(n_entity) < 0.7                              # the rest keep K independent draws (hubs)
        adj_e[low] = np.take_along_axis(pool_e, pick, 1)[low]
        adj_r[low] = np.take_along_axis(pool_r, pick, 1)[low]
    force_enc = bool(rng.random() < 0.5)      # the packed-tile kernel also below its auto thresholds
    what += f" repeats={repeats} force_enc={force_enc}"
    adj_e[:3] = 0                                                    # entities absent from the KG
    adj_r[:3] = 0
    params = init_params(args, n_user, n_entity, nR, seed=62000 + i, random_agg_bias=True)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=63000 + i)
    users = rng.integers(0, n_user, B)
    items = rng.integers(0, n_entity, B)
    outs = []
    for fused in (True, False):
        model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, params=params, device="cuda:0", fused=fused, table_dtype=tdt)
        if force_enc and fused:
            model.dedup = True
        dev = model.device
        u_d, i_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
        if feed == "pairs":
            mem = [[torch.from_numpy(x).to(dev) for x in lst] for lst in synth.memories_for(uts, users)]
            out = model.forward_device(u_d, i_d, *mem)
        else:
            model.group_min_pairs_per_user = 0 if feed == "users_grouped" else 10 ** 9
            out = model.forward_users(u_d, i_d, torch.from_numpy(uts).to(dev))
        torch.cuda.synchronize()
        outs.append((out.scores.cpu().numpy(), out.user_o.cpu().numpy(), out.item_embeddings.cpu().numpy()))
        del model, out
        torch.cuda.empty_cache()
    for name, a_, b_ in zip(("scores", "user_o", "item_embeddings"), outs[0], outs[1]):
        assert np.isfinite(a_).all() and np.isfinite(b_).all(), f"{name} not finite, {what}"
        assert_close(a_, b_, f"{name}: fused vs per-level, {what}", rtol=2e-5, atol=2e-6)
