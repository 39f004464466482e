"""-m gpu: the training step (mvin_amd/training.py) against oracle/train_ref.py: loss value,
every parameter gradient, and a short Adam trajectory."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from oracle import train_ref

pytestmark = pytest.mark.gpu

SHAPES = {
    "d8k3h2m1p2": dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=6),
    "d16k4h2m2p1": dict(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=2, p_hop=1, n_memory=8, batch_size=5),
    "d16k8h1m1p1": dict(dim=16, neighbor_sample_size=8, h_hop=1, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=7),
    "d64k8h2m1p2": dict(dim=64, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=16, batch_size=9),
    "d12k5h3m1p0": dict(dim=12, neighbor_sample_size=5, h_hop=3, n_mix_hop=1, p_hop=0, n_memory=4, batch_size=4),
}


def build(shape, ablation="all", seed=70, **extra):
    from mvin_amd.model import MVIN
    args = make_args(ablation=ablation, l2_weight=1e-3, l2_agg_weight=1e-4, lr=1e-2, **SHAPES[shape], **extra)
    case = synth.small_case(args, n_user=12, n_entity=120, n_relation=5, seed=seed, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=seed + 1, random_agg_bias=True)
    labels = (np.arange(args.batch_size) % 2).astype(np.float32)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0")
    return args, case, params, labels, model


def dev_feed(model, case, labels):
    dev = model.device
    return (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev), torch.from_numpy(labels).to(dev),
            [torch.from_numpy(m).to(dev) for m in case.memories_h], [torch.from_numpy(m).to(dev) for m in case.memories_r],
            [torch.from_numpy(m).to(dev) for m in case.memories_t])


def check_grads(got, ref, loss, ref_loss):
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-6, (loss, ref_loss)
    for name, g in ref.items():
        assert name in got, f"no gradient for {name}"
        scale = max(np.abs(g).max(), 1e-8)
        err = np.abs(got[name] - g).max()
        assert err <= 2e-4 * scale + 1e-7, f"{name}: max abs err {err:.3e} vs scale {scale:.3e}"
    for name, g in got.items():
        if name not in ref:
            assert not np.any(g), f"{name} has a gradient but the reference has none"


@pytest.mark.parametrize("shape,ablation", [("d8k3h2m1p2", "all"), ("d16k4h2m2p1", "all"), ("d16k8h1m1p1", "all"),
                                            ("d64k8h2m1p2", "all"), ("d8k3h2m1p2", "no_kg_eh_uo"),
                                            ("d8k3h2m1p2", "no_uo"), ("d8k3h2m1p2", "no_uor"),
                                            ("d8k3h2m1p2", "ho_only"), ("d8k3h2m1p2", "ho_only_uo_kg_eh"),
                                            ("d8k3h2m1p2", "no_ps_o_ft"), ("d8k3h2m1p2", "ps_only"),
                                            ("d12k5h3m1p0", "all"), ("d16k4h2m2p1", "no_uo_ho_only")])
def test_loss_and_every_gradient(shape, ablation, hip_lib):
    from mvin_amd.training import Trainer
    args, case, params, labels, model = build(shape, ablation)
    tr = Trainer(model)
    loss = tr.step(*dev_feed(model, case, labels), apply=False)
    torch.cuda.synchronize()
    ref_loss, ref_grads, _, _ = train_ref.loss_and_grads(args, params, case.adj_entity, case.adj_relation, case.users,
                                                         case.items, labels, case.memories_h, case.memories_r,
                                                         case.memories_t)
    check_grads(tr.grads_by_reference_name(), ref_grads, loss, ref_loss)


def test_adam_trajectory_matches_reference(hip_lib):
    args, case, params, labels, model = build("d8k3h2m1p2")
    feed = {model.user_indices: case.users, model.item_indices: case.items, model.labels: labels}
    for i in range(len(case.memories_h)):
        feed[model.memories_h[i]], feed[model.memories_r[i]], feed[model.memories_t[i]] = \
            case.memories_h[i], case.memories_r[i], case.memories_t[i]
    ref_p = {k: np.array(v, dtype=np.float32) for k, v in params.items()}
    opt = train_ref.AdamRef(ref_p, lr=args.lr)
    losses, ref_losses = [], []
    for _ in range(4):
        _, loss = model.train(None, feed)           # the reference's run wrapper (model.py:416-417)
        losses.append(loss)
        rl, rg, _, _ = train_ref.loss_and_grads(args, ref_p, case.adj_entity, case.adj_relation, case.users, case.items,
                                                labels, case.memories_h, case.memories_r, case.memories_t)
        ref_losses.append(rl)
        ref_p = opt.step(ref_p, rg)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4, atol=1e-6)
    assert losses[-1] < losses[0]
    got = model.parameters_dict()
    for k in ("entity_emb_matrix", "relation_emb_KGE_matrix", "agg_0_0_weights", "transfer_matrix_2", "user_mlp_matrix"):
        np.testing.assert_allclose(got[k], ref_p[k], rtol=0, atol=5e-4 * max(1.0, np.abs(ref_p[k]).max()))
    # scoring after training still matches the oracle evaluated at the reference's updated parameters
    from oracle import mirror_fp32
    _, s = model.get_scores(None, feed)
    ref = mirror_fp32.forward(args, ref_p, case.adj_entity, case.adj_relation, case.users, case.items, case.memories_h,
                              case.memories_r, case.memories_t)
    np.testing.assert_allclose(s, ref.scores_normalized.numpy(), atol=2e-3)
