"""-m gpu: the HIP path against the committed golden fixtures (inputs + expected outputs)."""
import glob
import os

import numpy as np
import pytest
import torch

from parity import ATOL, RTOL, assert_close
from test_oracle_golden import GOLDEN, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_matches_golden(path, hip_lib):
    from mvin_amd.model import MVIN
    z, args, params, (mh, mr, mt) = load_fixture(path)
    model = MVIN(args, int(z["n_user"]), int(z["n_entity"]), int(z["n_relation"]), z["adj_entity"],
                 z["adj_relation"], params=params, device="cuda:0")
    feed = {model.user_indices: z["users"], model.item_indices: z["items"],
            model.labels: np.ones(len(z["users"]), np.float32)}
    for i in range(len(mh)):
        feed[model.memories_h[i]], feed[model.memories_r[i]], feed[model.memories_t[i]] = mh[i], mr[i], mt[i]
    items, sig = model.get_scores(None, feed)
    np.testing.assert_array_equal(items, z["items"])
    assert_close(sig, z["scores_normalized"], "sigmoid scores vs golden")
    dev = model.device
    out = model.forward_device(torch.from_numpy(z["users"]).to(dev), torch.from_numpy(z["items"]).to(dev),
                               [torch.from_numpy(m).to(dev) for m in mh],
                               [torch.from_numpy(m).to(dev) for m in mr],
                               [torch.from_numpy(m).to(dev) for m in mt], want_probs=True)
    assert_close(out.scores.cpu().numpy(), z["scores_fp32"], "scores vs golden fp32")
    assert_close(out.scores.cpu().numpy(), z["scores_fp64"], "scores vs golden fp64", rtol=RTOL, atol=2 * ATOL)
    assert_close(out.user_o.cpu().numpy(), z["user_o"], "user_o")
    assert_close(out.item_embeddings.cpu().numpy(), z["item_embeddings"], "item_embeddings")
    for i, p in enumerate(out.importance_list):
        if p is not None:
            assert_close(p.cpu().numpy(), z[f"importance_{i}"], f"importance_{i}")
    # id expansion is integer work: bit-exact
    L = args.n_mix_hop * args.h_hop
    ents, rels = model.get_neighbors(torch.from_numpy(z["items"]).to(dev), levels=L)
    for i in range(L + 1):
        np.testing.assert_array_equal(ents[i].cpu().numpy(), z[f"entities_{i}"])
    for i in range(L):
        np.testing.assert_array_equal(rels[i].cpu().numpy(), z[f"relations_{i}"])
