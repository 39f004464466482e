"""-m gpu: mvin_l2_tail_fwd -- level-0 projection, both hop-0 aggregators, the mix-hop combiner and the score of a depth-2 tree
(model.py:270-317, aggregators.py:108-116, model.py:158-159) -- directly against a float64 evaluation of those formulas: the
barrier-free dim-64 kernel (mvin_tail_flash.hip), the tile-image kernel it replaces at dim 64 (MVIN_TAIL_FLASH=0 in a subprocess)
and at dims 16 / 32, ragged batch sizes, int32 / int64 ids, an id outside the table."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from parity import assert_close

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, D, n_entity, seed, idt=torch.int64):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5      # noqa: E731
    E = rnd(n_entity, D)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    q, uo, n0, n1 = rnd(B, D), rnd(B, D), rnd(B, D), rnd(B, D)
    W0, A0, A1, Wmix = rnd(D, D) * 0.4, rnd(D, D) * 0.4, rnd(D, D) * 0.4, rnd(3 * D, D) * 0.3
    b0, a0, a1, bmix = rnd(D), rnd(D), rnd(D), rnd(D)
    return E, items, q, uo, n0, n1, W0, b0, A0, a0, A1, a1, Wmix, bmix


def _reference(E, items, q, uo, n0, n1, W0, b0, A0, a0, A1, a1, Wmix, bmix):
    d = lambda t: t.double()      # noqa: E731
    ev0 = (d(E)[items.long()] + d(q)) @ d(W0) + d(b0)
    out0 = torch.relu((ev0 + d(n0)) @ d(A0) + d(a0))
    out2 = torch.relu((out0 + d(n1)) @ d(A1) + d(a1))
    item = torch.cat([ev0, out0, out2], dim=1) @ d(Wmix) + d(bmix)
    s = (d(uo) * item).sum(1)
    return item, s, torch.sigmoid(s)


@pytest.mark.parametrize("idt", [torch.int64, torch.int32], ids=["i64", "i32"])
@pytest.mark.parametrize("B", [1, 15, 16, 17, 33, 500, 4099, 70000])
@pytest.mark.parametrize("D", [64, 32, 16])
def test_tail_against_float64(D, B, idt, hip_lib):
    from mvin_amd import ops
    if idt == torch.int32 and B not in (17, 4099):
        pytest.skip("int32 ids: two sizes")
    args = _inputs(B, D, 3000, seed=D + B, idt=idt)
    item, s, sg = ops.l2_tail(*args)
    again = ops.l2_tail(*args)
    torch.cuda.synchronize()
    assert torch.equal(item, again[0]) and torch.equal(s, again[1])
    ri, rs, rg = _reference(*args)
    assert_close(item.cpu().numpy(), ri.cpu().numpy(), "item embeddings vs float64", rtol=1e-5, atol=2e-6)
    assert_close(s.cpu().numpy(), rs.cpu().numpy(), "scores vs float64", rtol=1e-5, atol=4e-6)
    assert_close(sg.cpu().numpy(), rg.cpu().numpy(), "sigmoid scores vs float64", rtol=1e-5, atol=1e-6)


def test_out_of_range_item_id_is_clamped(hip_lib):
    from mvin_amd import ops
    args = list(_inputs(100, 64, 500, seed=3))
    bad = args[1].clone()
    bad[::9] = 500 + 77
    args[1][::9] = 499
    x = ops.l2_tail(*args)
    args[1] = bad
    y = ops.l2_tail(*args)
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])


_CHILD = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from test_gpu_tail import _inputs, _reference
from mvin_amd import ops
args = _inputs(5000, 64, 3000, seed=11)
item, s, sg = ops.l2_tail(*args)
ri, rs, rg = _reference(*args)
assert float((item.double() - ri).abs().max()) < 1e-5 and float((s.double() - rs).abs().max()) < 2e-5
torch.save((item.cpu(), s.cpu()), sys.argv[1])
"""


def test_tile_image_kernel_at_dim_64_still_agrees(hip_lib, tmp_path):
    """MVIN_TAIL_FLASH=0 (read once per process): the kernel dim 64 took before, on the same inputs, in a subprocess."""
    from mvin_amd import ops
    out = tmp_path / "old.pt"
    env = dict(os.environ, MVIN_TAIL_FLASH="0")
    r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, ROOT), str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    old_item, old_s = torch.load(out)
    item, s, _ = ops.l2_tail(*_inputs(5000, 64, 3000, seed=11))
    assert_close(item.cpu().numpy(), old_item.numpy(), "flash tail vs tile-image kernel", rtol=1e-5, atol=2e-6)
    assert_close(s.cpu().numpy(), old_s.numpy(), "scores, flash tail vs tile-image kernel", rtol=1e-5, atol=4e-6)
