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


@pytest.mark.parametrize("shape", ["d8k3h2m1p2", "d64k8h2m1p2"])
def test_item_gradient_of_the_relation_projection_both_ways(shape, hip_lib):
    """dE[item] of V = E[item] . R_KGE[r]: added inside mvin_key_addressing_bwd_reg at small batches, by a separate
    product + scatter-add above Trainer.item_grad_in_kernel_max_batch -- both against the oracle."""
    from mvin_amd.training import Trainer
    for max_batch in (0, 1 << 20):
        args, case, params, labels, model = build(shape)
        tr = Trainer(model)
        tr.item_grad_in_kernel_max_batch = max_batch
        loss = tr.step(*dev_feed(model, case, labels), apply=False)
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


@pytest.mark.parametrize("shape,ablation", [("d8k3h2m1p2", "all"), ("d16k4h2m2p1", "all"), ("d64k8h2m1p2", "all"),
                                            ("d8k3h2m1p2", "ps_only"), ("d8k3h2m1p2", "no_uor"),
                                            ("d12k5h3m1p0", "all")])
def test_graphed_step_equals_eager_step_and_reference(shape, ablation, hip_lib):
    """training.GraphedTrainer: the step captured once as a hipGraph and replayed on a DIFFERENT batch every step
    follows the eager Trainer (same kernels; float atomics reorder sums) and oracle/train_ref.py."""
    from mvin_amd.training import GraphedTrainer, Trainer
    args, case, params, labels, model_e = build(shape, ablation)
    _, _, _, _, model_g = build(shape, ablation)
    tr_e, tr_g = Trainer(model_e), Trainer(model_g)
    gt = GraphedTrainer(tr_g, args.batch_size, ids_dtype=torch.from_numpy(case.users).dtype)
    # capture and its apply=False warm-up must leave parameters, moments and the step counter untouched
    assert tr_g.t == 0 and not torch.any(tr_g._m) and not torch.any(tr_g._v)
    for k, v in model_e.parameters_dict().items():
        np.testing.assert_array_equal(model_g.parameters_dict()[k], v, err_msg=k)
    ref_p = {k: np.array(v, dtype=np.float32) for k, v in params.items()}
    opt = train_ref.AdamRef(ref_p, lr=args.lr)
    rng = np.random.default_rng(5)
    le, lg, lr_ = [], [], []
    for _ in range(4):
        perm = rng.permutation(args.batch_size)
        sub = lambda x: np.ascontiguousarray(x[perm])
        c_u, c_i, c_l = sub(case.users), sub(case.items), sub(labels)
        mh, mr, mt = [sub(x) for x in case.memories_h], [sub(x) for x in case.memories_r], [sub(x) for x in case.memories_t]
        dev = model_e.device
        feed = (torch.from_numpy(c_u).to(dev), torch.from_numpy(c_i).to(dev), torch.from_numpy(c_l).to(dev),
                [torch.from_numpy(x).to(dev) for x in mh], [torch.from_numpy(x).to(dev) for x in mr],
                [torch.from_numpy(x).to(dev) for x in mt])
        le.append(tr_e.step(*feed))
        lg.append(float(gt.step(*feed).item()))
        rl, rg, _, _ = train_ref.loss_and_grads(args, ref_p, case.adj_entity, case.adj_relation, c_u, c_i, c_l, mh, mr, mt)
        lr_.append(rl)
        ref_p = opt.step(ref_p, rg)
    assert tr_g.t == tr_e.t == 4
    np.testing.assert_allclose(lg, le, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(lg, lr_, rtol=2e-4, atol=1e-6)
    pe, pg = model_e.parameters_dict(), model_g.parameters_dict()
    for k in pe:
        # Adam divides by sqrt(v): where a gradient is ~0 the reordering of float atomics can flip a whole step of
        # size lr, so the trajectories are compared at a few steps of slack, as the eager test does
        np.testing.assert_allclose(pg[k], pe[k], rtol=0, atol=5e-4 * max(1.0, np.abs(pe[k]).max()), err_msg=k)
    # scoring after the replays uses tables of the NEW weights (the graph leaves the model invalidated)
    feed_d = {model_g.user_indices: case.users, model_g.item_indices: case.items, model_g.labels: labels}
    feed_e = {model_e.user_indices: case.users, model_e.item_indices: case.items, model_e.labels: labels}
    for i in range(len(case.memories_h)):
        for f, mdl in ((feed_d, model_g), (feed_e, model_e)):
            f[mdl.memories_h[i]], f[mdl.memories_r[i]], f[mdl.memories_t[i]] = \
                case.memories_h[i], case.memories_r[i], case.memories_t[i]
    np.testing.assert_allclose(model_g.get_scores(None, feed_d)[1], model_e.get_scores(None, feed_e)[1], atol=2e-3)


def test_graphed_epoch_through_the_harness(hip_lib):
    """harness.train_epoch_device(graph=True) == the eager epoch on the same shuffled data."""
    from mvin_amd import harness
    args, case, params, labels, model_e = build("d8k3h2m1p2")
    _, _, _, _, model_g = build("d8k3h2m1p2")
    uts = synth.ripple_sets(case.n_user, case.n_entity, case.n_relation, args.p_hop, args.n_memory, seed=3)
    rng = np.random.default_rng(11)
    data = np.stack([rng.integers(0, case.n_user, 40), rng.integers(0, 60, 40), rng.integers(0, 2, 40)], axis=1)
    a = harness.train_epoch_device(harness.DeviceFeeder(model_e, uts), data.copy(), args.batch_size,
                                   rng=np.random.default_rng(1))
    b = harness.train_epoch_device(harness.DeviceFeeder(model_g, uts), data.copy(), args.batch_size,
                                   rng=np.random.default_rng(1), graph=True)
    assert len(a) == len(b) == 40 // args.batch_size
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-7)
    # a captured step must not outlive the adjacency it was captured on
    gt = model_g._graphed_trainer
    model_g.set_adjacency(case.adj_entity.copy(), case.adj_relation.copy())
    with pytest.raises(RuntimeError, match="GraphedTrainer"):
        gt.replay()


def test_train_wrapper_replays_a_captured_step_when_the_batch_size_repeats(hip_lib, monkeypatch):
    """MVIN.train (the reference's run wrapper, model.py:416-417): from the second step of a batch size on the step is one
    hipGraph replay; a sequence that mixes batch sizes gives the losses and parameters of the all-eager twin."""
    args, case, params, labels, model = build("d8k3h2m1p2")
    monkeypatch.setenv("MVIN_TRAIN_GRAPH", "0")
    _, _, _, _, twin = build("d8k3h2m1p2")

    def feed_of(m, n):
        f = {m.user_indices: case.users[:n], m.item_indices: case.items[:n], m.labels: labels[:n]}
        for i in range(len(case.memories_h)):
            f[m.memories_h[i]], f[m.memories_r[i]], f[m.memories_t[i]] = \
                case.memories_h[i][:n], case.memories_r[i][:n], case.memories_t[i][:n]
        return f
    B = len(case.users)
    sizes = [B, B, B, B - 1, B, B, B - 1, B - 1]
    ref_losses = [twin.train(None, feed_of(twin, n))[1] for n in sizes]
    assert not twin._train_graphs
    monkeypatch.setenv("MVIN_TRAIN_GRAPH", "1")
    losses = [model.train(None, feed_of(model, n))[1] for n in sizes]
    assert set(model._train_graphs) == {B, B - 1}          # both sizes were captured once they repeated
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-5, atol=1e-7)
    a, b = model.parameters_dict(), twin.parameters_dict()
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=1e-6, err_msg=k)
