"""-m gpu: hipGraph replay of the scoring path gives bit-identical scores to eager launches."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params

pytestmark = pytest.mark.gpu


def test_graph_replay_matches_eager(hip_lib):
    from mvin_amd.graph import GraphedScorer
    from mvin_amd.model import MVIN
    args = make_args(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=128)
    dev = torch.device("cuda:0")
    nE = 3000
    adj_e, adj_r = synth.uniform_adjacency(nE, 9, 32, seed=1)
    params = init_params(args, 200, nE, 9, seed=2)
    model = MVIN(args, 200, nE, 9, adj_e, adj_r, params=params, device=dev)
    scorer = GraphedScorer(model, 128)
    rng = np.random.default_rng(3)
    for trial in range(3):     # new inputs every replay
        users = torch.from_numpy(rng.integers(0, 200, 128)).to(dev)
        items = torch.from_numpy(rng.integers(0, nE, 128)).to(dev)
        mh = [torch.from_numpy(rng.integers(0, nE, (128, 64)).astype(np.int32)).to(dev) for _ in range(2)]
        mr = [torch.from_numpy(rng.integers(0, 9, (128, 64)).astype(np.int32)).to(dev) for _ in range(2)]
        mt = [torch.from_numpy(rng.integers(0, nE, (128, 64)).astype(np.int32)).to(dev) for _ in range(2)]
        got = scorer(users, items, mh, mr, mt).scores.clone()
        ref = model.forward_device(users, items, mh, mr, mt).scores
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


def test_stale_graph_is_refused(hip_lib):
    """A captured graph reads derived tables that an optimizer step / new adjacency replaces."""
    from mvin_amd.graph import GraphedScorer
    from mvin_amd.model import MVIN
    args = make_args(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=16)
    adj_e, adj_r = synth.uniform_adjacency(100, 5, 4, seed=1)
    model = MVIN(args, 20, 100, 5, adj_e, adj_r, device="cuda:0", seed=2, hoist=True)
    scorer = GraphedScorer(model, 16)
    scorer.replay()
    model.set_adjacency(adj_e, adj_r)
    with pytest.raises(RuntimeError, match="new GraphedScorer"):
        scorer.replay()
    scorer = GraphedScorer(model, 16)      # entity tables are rebuilt in the warm-up, outside the capture
    a = scorer.replay().scores.clone()
    b = model.forward_device(scorer.users, scorer.items, scorer.mh, scorer.mr, scorer.mt).scores
    assert torch.equal(a, b)
