"""-m gpu: the harness counterpart on the real model: host-feed and device-feed evaluations
agree, and the case-study dump has the reference's layout."""
import numpy as np
import pytest

from mvin_amd import harness, synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params

pytestmark = pytest.mark.gpu


def build():
    from mvin_amd.model import MVIN
    args = make_args(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=8, batch_size=32)
    n_user, n_entity, n_relation, n_item = 30, 400, 6, 60
    rng = np.random.default_rng(7)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, n_relation, 4, seed=8)
    uts = synth.ripple_sets(n_user, n_entity, n_relation, 2, 8, seed=9)
    params = init_params(args, n_user, n_entity, n_relation, seed=10, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, n_relation, adj_e, adj_r, params=params, device="cuda:0")
    data = np.stack([rng.integers(0, n_user, 700), rng.integers(0, n_item, 700), rng.integers(0, 2, 700)], axis=1)
    return args, model, uts, data, n_item


def test_ctr_eval_host_and_device_feeds_agree(hip_lib):
    args, model, uts, data, _ = build()
    a = harness.ctr_eval(args, model, data, uts, 32)
    b = harness.ctr_eval_device(harness.DeviceFeeder(model, uts), data, 32)
    assert len(a[0]) == 700 // 32
    np.testing.assert_allclose(a[0], b[0], atol=1e-6)
    np.testing.assert_allclose(a[3:], b[3:], atol=1e-6)


def test_topk_eval_host_and_device_feeds_agree(hip_lib):
    args, model, uts, data, n_item = build()
    users, tr, ev, te, item_set, k_list = harness.topk_settings(data[:450], data[450:570], data[570:], n_item, user_num=8)
    k_list = [1, 2, 5, 10, 25]
    a = harness.topk_eval(args, uts, model, users, tr, ev, te, item_set, k_list, 32, mode="eval")
    b = harness.topk_eval_device(harness.DeviceFeeder(model, uts), users, tr, ev, te, item_set, k_list, 32, mode="eval")
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9)


def test_case_study_dump_layout(hip_lib, tmp_path):
    args, model, uts, data, n_item = build()
    hist = harness.get_user_record(data)
    path = tmp_path / "case.log"
    harness.ctr_eval_case_study(args, model, data[:64], uts, hist, {"3": "Entity Three"}, {"0": "rel zero"},
                                set(range(30)), set(range(n_item)), 32, str(path))
    txt = path.read_text().splitlines()
    assert txt[1] == " case_study "
    assert sum(line.startswith("user_indices = ") for line in txt) == 64
    first = txt.index(next(l for l in txt if l.startswith("user_indices = ")))
    assert txt[first + 1].endswith("first_layer  " + "*" * 20)
    assert txt[first + 2].startswith("et_index 0 = ") and txt[first + 3].startswith("rela_index 0 = ")
    att = [l for l in txt if l.startswith("er rela pair 0 = ")][0]
    assert "att = 0." in att or "att = 1.0" in att
    # second layer blocks: K entities per pair, K attention weights each
    assert sum(l.startswith("er rela pair 1 = ") for l in txt) == 64 * 4
