"""-m gpu: KG -> GPU-built inputs -> training epochs -> evaluation; the model must learn the
synthetic signal (loss falls, eval AUC well above chance)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_train_and_eval_on_synthetic_signal(hip_lib, monkeypatch):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import run_synthetic
    monkeypatch.setattr(sys, "argv", ["run_synthetic.py", "--epochs", "5"])
    out = run_synthetic.main()
    assert out["loss"][-1] < out["loss"][0] - 0.05          # it trains
    assert max(out["auc"]) > 0.54                           # and generalises to held-out pairs (chance = 0.5)
