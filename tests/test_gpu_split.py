"""-m gpu: the role-split fused kernel (mvin_fused_split.hip) -- the kernel bench.py times -- against the
oracles.  It only runs when no attention outputs are requested (want_probs=False), which the other parity
tests do request, so every template instance (D x K x table dtype x tree depth x ablation switch) is checked
here, with mvin_gather_attn_l2_variant() asserting that this is the kernel the call took."""
import os

import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from oracle import mirror_fp32

from parity import assert_close, run_hip, run_oracles

pytestmark = pytest.mark.gpu

# (D, K) pairs the role-split kernel is instantiated for
DK = [(32, 16), (32, 32), (32, 64), (32, 128), (64, 32), (64, 64), (64, 128), (128, 32), (128, 64), (128, 128)]


def _variant(D, K):
    """(32, 16) goes to the wave-per-parent kernel of mvin_fused_d32.hip (variant 4) unless MVIN_L2_D32=0 -- under which
    tests/test_gpu_d32.py re-runs this file's D32K16 cases in a subprocess, so the pipeline's <32, 16> instance stays covered."""
    return 4 if (D, K) == (32, 16) and os.environ.get("MVIN_L2_D32", "1") != "0" else 2


def _shape(D, K, H=2, B=None):
    if B is None:
        B = max(2, min(37, 4096 // (K * K)))           # what the CPU oracles finish in seconds
    return dict(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=2, n_memory=8, batch_size=B)


def _check(args, case, params, table_dtype="f32", oracle_params=None, rtol=1e-5, atol=1e-6):
    n_parents = case.users.shape[0] * args.neighbor_sample_size ** (args.h_hop - 2)
    assert ops.gather_attn_l2_variant(args.dim, args.neighbor_sample_size, n_parents, case.n_entity, False) == _variant(args.dim, args.neighbor_sample_size)
    assert ops.gather_attn_l2_variant(args.dim, args.neighbor_sample_size, n_parents, case.n_entity, True) == 1
    _, out = run_hip(args, case, params, want_probs=False, table_dtype=table_dtype)
    m, e = run_oracles(args, case, oracle_params or params)
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), "role-split scores vs fp32 mirror", rtol=rtol, atol=atol)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), "item_embeddings", rtol=rtol, atol=atol)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"
    return out


@pytest.mark.parametrize("table", ["f32", "bf16"])
@pytest.mark.parametrize("dk", DK, ids=lambda dk: "D%dK%d" % dk)
def test_split_kernel_vs_oracles(dk, table, hip_lib):
    D, K = dk
    args = make_args(**_shape(D, K))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=41 + D + K, zero_rows=4)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=43, random_agg_bias=True)
    if table == "bf16":
        rounded = dict(params, entity_emb_matrix=torch.from_numpy(params["entity_emb_matrix"]).to(torch.bfloat16).float().numpy())
        _check(args, case, params, "bf16", oracle_params=rounded)
    else:
        _check(args, case, params)


@pytest.mark.parametrize("dk", [(32, 16), (64, 32)], ids=lambda dk: "D%dK%d" % dk)
def test_split_kernel_depth3(dk, hip_lib):
    """h_hop = 3: K parents per pair, the query row shared by the K parents of a pair."""
    D, K = dk
    args = make_args(**_shape(D, K, H=3, B=3))
    case = synth.small_case(args, n_user=8, n_entity=700, n_relation=5, seed=51, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=52, random_agg_bias=True)
    _check(args, case, params)


@pytest.mark.parametrize("ablation", ["no_uor", "no_uo", "no_uor_and_no_kg_eh_uo", "no_uo_and_no_kg_eh_uo"])
@pytest.mark.parametrize("dk", [(32, 16), (64, 32), (128, 32)], ids=lambda dk: "D%dK%d" % dk)
def test_split_kernel_without_attention_or_projection(dk, ablation, hip_lib):
    """The switches that change what the kernel computes: User_orient_rela off = uniform neighbor weights
    (aggregators.py:124-127), User_orient off = no query projection (model.py:270-283)."""
    D, K = dk
    args = make_args(ablation=ablation, **_shape(D, K, B=11))
    case = synth.small_case(args, n_user=8, n_entity=600, n_relation=6, seed=61)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=62, random_agg_bias=True)
    _check(args, case, params)


@pytest.mark.parametrize("dk", [(32, 16), (64, 32)], ids=lambda dk: "D%dK%d" % dk)
@pytest.mark.parametrize("B", [1, 2, 255, 4097])
def test_split_kernel_matches_symmetric_kernel_at_ragged_sizes(dk, B, hip_lib):
    """Parent counts around the kernel's work split (one parent, fewer parents than workgroups, not a multiple
    of the workgroup count): the two fused kernels are independent programs and must agree to fp32 round-off;
    the symmetric one is the kernel every want_probs parity test pins to the oracle."""
    D, K = dk
    args = make_args(**_shape(D, K, B=B))
    case = synth.small_case(args, n_user=32, n_entity=2000, n_relation=9, seed=71 + B)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=72, random_agg_bias=True)
    assert ops.gather_attn_l2_variant(D, K, B, case.n_entity, False) == _variant(D, K)
    _, a = run_hip(args, case, params, want_probs=False)
    _, b = run_hip(args, case, params, want_probs=True)
    assert_close(a.scores.cpu().numpy(), b.scores.cpu().numpy(), "role-split vs symmetric fused kernel")
    # and a pair's score does not depend on where in the batch it sits
    if B > 4:
        sl = slice(B - 3, B)
        import copy
        c2 = copy.copy(case)
        for f in ("users", "items"):
            setattr(c2, f, getattr(case, f)[sl])
        for f in ("memories_h", "memories_r", "memories_t"):
            setattr(c2, f, [m[sl] for m in getattr(case, f)])
        a2 = run_hip(make_args(**_shape(D, K, B=3)), c2, params, want_probs=False)[1]
        assert torch.equal(a2.scores, a.scores[sl])


@pytest.mark.parametrize("name", ["C2", "C3", "C4"])
def test_split_kernel_dataset_sized(name, hip_lib):
    """BASELINE configs at dataset-sized tables: oracle on a sample, and batch independence."""
    from test_gpu_properties import CONFIGS, setup
    args, case, params, model = setup(name)
    dev = model.device
    feed = lambda sl: (torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
                       [torch.from_numpy(m[sl]).to(dev) for m in case.memories_h],
                       [torch.from_numpy(m[sl]).to(dev) for m in case.memories_r],
                       [torch.from_numpy(m[sl]).to(dev) for m in case.memories_t])
    B = case.users.shape[0]
    assert ops.gather_attn_l2_variant(args.dim, args.neighbor_sample_size, B, case.n_entity, False) == _variant(args.dim, args.neighbor_sample_size)
    out = model.forward_device(*feed(slice(None)), want_probs=False)
    n = {"C2": 64, "C3": 32, "C4": 8}[name]
    sl = slice(0, n)
    sargs = make_args(**dict(vars(args), batch_size=n))
    ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, case.users[sl], case.items[sl],
                              [m[sl] for m in case.memories_h], [m[sl] for m in case.memories_r],
                              [m[sl] for m in case.memories_t])
    assert_close(out.scores[sl].cpu().numpy(), ref.scores.numpy(), f"{name} role-split scores vs fp32 mirror")
    out2 = model.forward_device(*feed(slice(5, 5 + n)), want_probs=False)
    assert torch.equal(out2.scores, out.scores[5:5 + n])
    del CONFIGS


@pytest.mark.parametrize("table", ["f32", "bf16"])
@pytest.mark.parametrize("dk", [(16, 8), (32, 16), (64, 32), (128, 5)], ids=lambda dk: "D%dK%d" % dk)
def test_gather_only_probe_reads_the_rows_of_the_fused_kernel(dk, table, hip_lib):
    """mvin_probe_gather_l2 (bench.py's empirical ceiling for the fused kernel's row gathers): per parent the sum of
    every element of its K child rows and K*K grandchild rows."""
    D, K = dk
    if table == "bf16" and D == 16:
        pytest.skip("32-byte rows")
    rng = np.random.default_rng(D + K)
    nE, P = 3000, 777
    E = torch.from_numpy(rng.normal(size=(nE, D)).astype(np.float32)).cuda()
    if table == "bf16":
        E = E.to(torch.bfloat16)
    adj = torch.from_numpy(rng.integers(0, nE, (nE, K)).astype(np.int32)).cuda()
    parents = torch.from_numpy(rng.integers(0, nE, P).astype(np.int32)).cuda()
    x1 = adj[parents.long()].long()                      # [P, K]
    y = adj[x1].long()                                   # [P, K, K]
    got = ops.probe_gather_l2(E, x1.to(torch.int32).contiguous(), y.to(torch.int32).contiguous(), K)
    torch.cuda.synchronize()
    Ed = E.double()
    ref = Ed[x1].sum((1, 2)) + Ed[y].sum((1, 2, 3))
    assert_close(got.cpu().numpy(), ref.cpu().numpy(), "probe sums", rtol=1e-4, atol=1e-2)
