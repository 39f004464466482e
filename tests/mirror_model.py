"""Test helper: the reference's run-wrapper surface (model.py:416-444) served by the CPU oracle
(oracle/mirror_fp32.py), so the harness counterpart can be driven by the oracle and by the HIP model
alike and both compared with the reference-produced fixture tests/golden/ref/harness.npz."""
import json
import os
import re

import numpy as np

from mvin_amd.config import make_args
from oracle import mirror_fp32

REF = os.path.join(os.path.dirname(__file__), "golden", "ref")


def load_harness_fixture():
    z = np.load(os.path.join(REF, "harness.npz"))
    args = make_args(**json.loads(str(z["args_json"])))
    params = {k[3:]: z[k] for k in z.files if k.startswith("p__")}
    return z, args, params, z["uts"]


class MirrorModel(object):
    def __init__(self, args, params, adj_entity, adj_relation):
        self.args, self.params, self.adj_e, self.adj_r = args, params, adj_entity, adj_relation
        P = max(1, args.p_hop)
        self.user_indices, self.item_indices, self.labels = "user_indices", "item_indices", "labels"
        self.memories_h = [f"memories_h_{i}" for i in range(P)]
        self.memories_r = [f"memories_r_{i}" for i in range(P)]
        self.memories_t = [f"memories_t_{i}" for i in range(P)]

    def _run(self, feed):
        P = len(self.memories_h)
        users, items = np.asarray(feed[self.user_indices]), np.asarray(feed[self.item_indices])
        mem = [[np.asarray(feed[m[i]]) for i in range(P)] for m in (self.memories_h, self.memories_r, self.memories_t)]
        a = make_args(**dict(vars(self.args), batch_size=len(items)))
        return mirror_fp32.forward(a, self.params, self.adj_e, self.adj_r, users, items, *mem)

    def get_scores(self, sess, feed):
        return np.asarray(feed[self.item_indices]), self._run(feed).scores_normalized.numpy()

    def eval(self, sess, feed):
        from sklearn.metrics import f1_score, roc_auc_score
        scores = self._run(feed).scores_normalized.numpy()
        labels = np.asarray(feed[self.labels])
        auc = roc_auc_score(y_true=labels, y_score=scores)
        pred = (scores >= 0.5).astype(np.float32)
        return auc, float(np.mean(pred == labels)), f1_score(y_true=labels, y_pred=pred)

    def eval_case_study(self, sess, feed):
        out = self._run(feed)
        imp = out.importance_list
        imp0 = imp[0].numpy() if imp and imp[0] is not None else None
        imp1 = imp[1].numpy() if len(imp) > 1 and imp[1] is not None else 0
        return (np.asarray(feed[self.user_indices]), np.asarray(feed[self.labels], dtype=np.float32), np.asarray(feed[self.item_indices]),
                [e.numpy() for e in out.entities], [r.numpy() for r in out.relations], imp0, imp1)


_ATT = re.compile(r"att = ([-+0-9.eE]+)")


def compare_case_study_text(got, want, atol):
    """The dump of util.py:59-127: identical line by line except for the printed attention weights,
    which must agree within ``atol``.  Returns the number of attention values compared."""
    g, w = got.splitlines(), want.splitlines()
    assert len(g) == len(w), (len(g), len(w))
    n = 0
    for i, (a, b) in enumerate(zip(g, w)):
        va, vb = _ATT.findall(a), _ATT.findall(b)
        assert _ATT.sub("att = #", a) == _ATT.sub("att = #", b), f"line {i}: {a!r} != {b!r}"
        assert len(va) == len(vb)
        for x, y in zip(va, vb):
            assert abs(float(x) - float(y)) <= atol, (i, x, y)
            n += 1
    return n
