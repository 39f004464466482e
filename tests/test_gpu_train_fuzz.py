"""-m gpu: seeded random sweep over the configuration space of the TRAINING step (mvin_amd/training.py: forward in
its `_ex` form, loss of model.py:378-412, backward, no update) against oracle/train_ref.py (torch autograd on the fp32
mirror): loss value and every parameter gradient.  The hand-picked cases of test_gpu_train.py pin each backward kernel;
this one looks for interactions (depth 3 with two mix blocks' worth of stages, one memory per hop, a batch of one,
presets without attention, dims that take the VALU kernels, ...).  wide_deep=False presets are not trainable (nor in
the reference: model.py:366-374 raises) and are skipped by the draw."""
import os

import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args
from mvin_amd.params import init_params
from oracle import train_ref

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("MVIN_TRAIN_FUZZ_CASES", "24"))     # a longer campaign: MVIN_TRAIN_FUZZ_CASES=500
OFFSET = int(os.environ.get("MVIN_TRAIN_FUZZ_OFFSET", "0"))               # first case number (fresh draws for a campaign)


def _draw(i):
    rng = np.random.default_rng(7000 + i)
    while True:
        abl = str(rng.choice(sorted(ABLATIONS)))
        if make_args(ablation=abl).wide_deep:
            break
    D = int(rng.choice([4, 8, 12, 16, 32, 64]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8]))
    H = int(rng.choice([1, 2, 2, 3]))
    M = int(rng.choice([1, 1, 2]))
    while K ** (H * M) > 600:
        if M > 1:
            M = 1
        elif H > 1:
            H -= 1
        else:
            K = 4
    P = int(rng.choice([0, 1, 2, 2, 3]))
    if P == 0 and not make_args(ablation=abl).PS_O_ft:
        P = 1                                       # no read at all: rejected at construction, as model.py:232 would fail
    Nm = int(rng.choice([1, 3, 4, 8, 16]))
    nR = int(rng.choice([1, 2, 5, 9]))
    B = int(rng.choice([1, 2, 5, 9, 17, 40]))
    n_user = int(rng.choice([1, 3, 12]))
    return dict(D=D, K=K, H=H, M=M, P=P, Nm=Nm, nR=nR, B=B, n_user=n_user, abl=abl)


@pytest.mark.parametrize("i", range(OFFSET, OFFSET + N_CASES))
def test_random_training_configuration(i, hip_lib):
    from mvin_amd.model import MVIN
    from mvin_amd.training import Trainer
    c = _draw(i)
    args = make_args(ablation=c["abl"], l2_weight=1e-3, l2_agg_weight=1e-4, lr=1e-2, dim=c["D"], neighbor_sample_size=c["K"],
                     h_hop=c["H"], n_mix_hop=c["M"], p_hop=c["P"], n_memory=c["Nm"], batch_size=c["B"])
    case = synth.small_case(args, n_user=c["n_user"], n_entity=90 + 13 * (i % 4), n_relation=c["nR"], seed=7100 + i, zero_rows=2)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=7200 + i, random_agg_bias=True)
    labels = (np.random.default_rng(7300 + i).random(c["B"]) < 0.5).astype(np.float32)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0")
    dev = model.device
    feed = (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev), torch.from_numpy(labels).to(dev),
            [torch.from_numpy(m).to(dev) for m in case.memories_h], [torch.from_numpy(m).to(dev) for m in case.memories_r],
            [torch.from_numpy(m).to(dev) for m in case.memories_t])
    tr = Trainer(model)
    loss = tr.step(*feed, apply=False)
    torch.cuda.synchronize()
    ref_loss, ref, _, _ = train_ref.loss_and_grads(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                                                   labels, case.memories_h, case.memories_r, case.memories_t)
    what = f"case {i} {c}"
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-6, (loss, ref_loss, what)
    got = tr.grads_by_reference_name()
    for name, g in ref.items():
        assert name in got, f"no gradient for {name}, {what}"
        scale = max(np.abs(g).max(), 1e-8)
        err = np.abs(got[name] - g).max()
        assert err <= 2e-4 * scale + 1e-7, f"{name}: max abs err {err:.3e} vs scale {scale:.3e}, {what}"
    for name, g in got.items():
        if name not in ref:
            assert not np.any(g), f"{name} has a gradient but the reference has none, {what}"


N_TRAJ = int(os.environ.get("MVIN_TRAJ_FUZZ_CASES", "8"))


@pytest.mark.parametrize("i", range(OFFSET, OFFSET + N_TRAJ))
def test_random_training_trajectory_eager_graphed_reference(i, hip_lib):
    """Three optimisation steps on a different batch each, for a random configuration: the eager Trainer, the step
    replayed as a hipGraph (training.GraphedTrainer) and oracle/train_ref.py (autograd + tf.train.AdamOptimizer's rule)
    must report the same losses; eager and graphed parameters stay together."""
    from mvin_amd.model import MVIN
    from mvin_amd.training import GraphedTrainer, Trainer
    c = _draw(20000 + i)
    B = max(c["B"], 2)
    args = make_args(ablation=c["abl"], l2_weight=1e-3, l2_agg_weight=1e-4, lr=5e-3, dim=c["D"], neighbor_sample_size=c["K"],
                     h_hop=c["H"], n_mix_hop=c["M"], p_hop=c["P"], n_memory=c["Nm"], batch_size=B)
    big = make_args(**dict(vars(args), batch_size=3 * B))
    case = synth.small_case(big, n_user=c["n_user"], n_entity=90 + 13 * (i % 4), n_relation=c["nR"], seed=27100 + i, zero_rows=2)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=27200 + i, random_agg_bias=True)
    labels = (np.random.default_rng(27300 + i).random(3 * B) < 0.5).astype(np.float32)
    mk = lambda: MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                      device="cuda:0")
    model_e, model_g = mk(), mk()
    dev = model_e.device
    tr_e, tr_g = Trainer(model_e), Trainer(model_g)
    gt = GraphedTrainer(tr_g, B, ids_dtype=torch.from_numpy(case.users).dtype)
    ref_p = {k: np.array(v, dtype=np.float32) for k, v in params.items()}
    opt = train_ref.AdamRef(ref_p, lr=args.lr)
    le, lg, lr_ = [], [], []
    what = f"case {i} {c}"
    for s in range(3):
        sl = slice(s * B, (s + 1) * B)
        cut = lambda x: np.ascontiguousarray(x[sl])
        c_u, c_i, c_l = cut(case.users), cut(case.items), cut(labels)
        mh, mr, mt = [cut(x) for x in case.memories_h], [cut(x) for x in case.memories_r], [cut(x) for x in case.memories_t]
        feed = (torch.from_numpy(c_u).to(dev), torch.from_numpy(c_i).to(dev), torch.from_numpy(c_l).to(dev),
                [torch.from_numpy(x).to(dev) for x in mh], [torch.from_numpy(x).to(dev) for x in mr],
                [torch.from_numpy(x).to(dev) for x in mt])
        le.append(tr_e.step(*feed))
        lg.append(float(gt.step(*feed).item()))
        rl, rg, _, _ = train_ref.loss_and_grads(args, ref_p, case.adj_entity, case.adj_relation, c_u, c_i, c_l, mh, mr, mt)
        lr_.append(rl)
        ref_p = opt.step(ref_p, rg)
    np.testing.assert_allclose(lg, le, rtol=5e-5, atol=1e-6, err_msg=what)
    np.testing.assert_allclose(le, lr_, rtol=1e-3, atol=1e-5, err_msg=what)
    pe, pg = model_e.parameters_dict(), model_g.parameters_dict()
    for k in pe:
        np.testing.assert_allclose(pg[k], pe[k], rtol=0, atol=5e-4 * max(1.0, np.abs(pe[k]).max()), err_msg=f"{k}, {what}")
