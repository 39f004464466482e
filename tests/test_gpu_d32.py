"""-m gpu: the wave-per-parent fused kernel for D = 32, K in {8, 16} (mvin_fused_d32.hip) -- BASELINE config C2
(MovieLens-1M, dim 32, fan-out 16, depth 2) -- against the oracles and against the symmetric fused kernel.  It runs
only when no attention outputs are requested, which the generic parity tests do request, so every template instance
is checked here, with mvin_gather_attn_l2_variant() asserting that this is the kernel the call took."""
import copy

import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import ABLATIONS, make_args
from mvin_amd.params import init_params
from oracle import mirror_fp32

from parity import assert_close, run_hip, run_oracles

pytestmark = pytest.mark.gpu

KS = [8, 16]


def _shape(K, H=2, M=1, B=None, P=2, Nm=8):
    if B is None:
        B = 37
    return dict(dim=32, neighbor_sample_size=K, h_hop=H, n_mix_hop=M, p_hop=P, n_memory=Nm, batch_size=B)


def _check(args, case, params, table_dtype="f32", oracle_params=None, rtol=1e-5, atol=1e-6):
    L = args.h_hop * args.n_mix_hop
    n_parents = case.users.shape[0] * args.neighbor_sample_size ** (L - 2)
    assert ops.gather_attn_l2_variant(32, args.neighbor_sample_size, n_parents, case.n_entity, False) == 4
    assert ops.gather_attn_l2_variant(32, args.neighbor_sample_size, n_parents, case.n_entity, True) == 1
    _, out = run_hip(args, case, params, want_probs=False, table_dtype=table_dtype)
    m, e = run_oracles(args, case, oracle_params or params)
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), "wave-per-parent scores vs fp32 mirror", rtol=rtol, atol=atol)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), "item_embeddings", rtol=rtol, atol=atol)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"
    return out


@pytest.mark.parametrize("table", ["f32", "bf16"])
@pytest.mark.parametrize("K", KS)
def test_d32_kernel_vs_oracles(K, table, hip_lib):
    args = make_args(**_shape(K))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=141 + K, zero_rows=4)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=143, random_agg_bias=True)
    if table == "bf16":
        rounded = dict(params, entity_emb_matrix=torch.from_numpy(params["entity_emb_matrix"]).to(torch.bfloat16).float().numpy())
        _check(args, case, params, "bf16", oracle_params=rounded)
    else:
        _check(args, case, params)


@pytest.mark.parametrize("K", [8, 16])
def test_d32_kernel_depth3_and_two_mix_blocks(K, hip_lib):
    """h_hop = 3 (K parents per pair share the pair's query row) and n_mix_hop = 2 with h_hop = 2 (depth 4: K^2 parents)."""
    for H, M, B in ((3, 1, 5), (2, 2, 3)):
        if K ** (H * M) > 5000:
            continue
        args = make_args(**_shape(K, H=H, M=M, B=B))
        case = synth.small_case(args, n_user=8, n_entity=700, n_relation=5, seed=151 + H, zero_rows=3)
        params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=152, random_agg_bias=True)
        _check(args, case, params)


@pytest.mark.parametrize("ablation", sorted(a for a in ABLATIONS if "wd" not in a))
def test_d32_kernel_every_ablation_at_the_shipped_shape(ablation, hip_lib):
    """dim 32, fan-out 16, h_hop 2, p_hop 2, n_memory 64 (BASELINE C2), every --ablation preset (the switches change
    what the kernel computes: uniform weights without User_orient_rela, no projection without User_orient)."""
    args = make_args(ablation=ablation, **_shape(16, B=21, Nm=64))
    case = synth.small_case(args, n_user=8, n_entity=600, n_relation=9, seed=161)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=162, random_agg_bias=True)
    if args.PS_only or args.h_hop < 2:
        pytest.skip("no tree aggregation in this preset")
    _check(args, case, params)


@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("B", [1, 3, 255, 4097, 20001])
def test_d32_kernel_matches_symmetric_kernel_at_ragged_sizes(K, B, hip_lib):
    """Parent counts around the kernel's work split (one parent, fewer parents than waves, not a multiple of the
    wave count, more than one grid-stride round): the two kernels are independent programs and must agree to fp32
    round-off; the symmetric one is the kernel every want_probs parity test pins to the oracle."""
    args = make_args(**_shape(K, B=B, Nm=4, P=1))
    case = synth.small_case(args, n_user=32, n_entity=2000, n_relation=9, seed=171 + B)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=172, random_agg_bias=True)
    assert ops.gather_attn_l2_variant(32, K, B, case.n_entity, False) == 4
    _, a = run_hip(args, case, params, want_probs=False)
    _, b = run_hip(args, case, params, want_probs=True)
    assert_close(a.scores.cpu().numpy(), b.scores.cpu().numpy(), "wave-per-parent (D = 32) vs symmetric fused kernel")
    if B > 4:       # a pair's score does not depend on where in the batch it sits
        sl = slice(B - 3, B)
        c2 = copy.copy(case)
        for f in ("users", "items"):
            setattr(c2, f, getattr(case, f)[sl])
        for f in ("memories_h", "memories_r", "memories_t"):
            setattr(c2, f, [m[sl] for m in getattr(case, f)])
        a2 = run_hip(make_args(**_shape(K, B=3, Nm=4, P=1)), c2, params, want_probs=False)[1]
        assert torch.equal(a2.scores, a.scores[sl])


def test_d32_kernel_dataset_sized_c2(hip_lib):
    """BASELINE config C2 at dataset-sized tables (MovieLens-1M shape): oracle on a sample + batch independence."""
    from mvin_amd.model import MVIN
    d = synth.DATASETS["MovieLens-1M"]
    B = 20000
    args = make_args(dataset="MovieLens-1M", dim=32, neighbor_sample_size=16, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"],
                     n_memory=d["n_memory"], batch_size=B)
    case = synth.dataset_case("MovieLens-1M", K=16, B=B, seed=5)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=6, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0")
    dev = model.device
    feed = lambda sl: (torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
                       [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_h],
                       [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_r],
                       [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_t])
    assert ops.gather_attn_l2_variant(32, 16, B, case.n_entity, False) == 4
    out = model.forward_device(*feed(slice(None)))
    n = 96
    sargs = make_args(**dict(vars(args), batch_size=n))
    sl = slice(0, n)
    ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, case.users[sl], case.items[sl],
                              [m[sl] for m in case.memories_h], [m[sl] for m in case.memories_r],
                              [m[sl] for m in case.memories_t])
    assert_close(out.scores[sl].cpu().numpy(), ref.scores.numpy(), "C2, dataset-sized: scores vs fp32 mirror")
    out2 = model.forward_device(*feed(slice(7, 7 + n)))
    assert torch.equal(out2.scores, out.scores[7:7 + n])


def test_pipeline_instance_for_d32_k16_still_passes(hip_lib):
    """MVIN_L2_D32=0 hands (D, K) = (32, 16) back to the role-split pipeline: its <32, 16> instance keeps its tests
    (the switch is read once per process, hence the subprocess)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_split.py"), "-q", "-x", "-k", "D32K16",
                        "-p", "no:cacheprovider"], env=dict(os.environ, MVIN_L2_D32="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
