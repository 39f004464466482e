"""-m gpu: the launches bench.py quotes its roofline on are CHECKED launches -- the 16 M-row (4.1 GB) table of the HBM
leg, and tables above 4 GiB, where the fused kernels leave the 32-bit buffer-offset addressing (fp32: MODE 2, bf16: MODE 0
of mvin_fused_split.hip).  Reference = bench.l2_reference_f64 (plain torch indexing in float64 on the GPU, written from
include/mvin_hip.h's formulas: model.py:295-305, aggregators.py:98-146), 256 parents sampled across each launch."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvin_amd import ops  # noqa: E402

pytestmark = pytest.mark.gpu


def test_hbm_leg_launch_is_verified(hip_lib):
    leg = bench.hbm_leg(torch.device("cuda:0"), 64, 32, "f32", iters=1, warmup=1)
    assert leg["outputs_finite"] and leg["verified"], leg
    enc = bench.hbm_leg(torch.device("cuda:0"), 64, 32, "f32", iters=1, warmup=1, encoded=True)     # the packed-tile kernel on the same table
    assert enc["verified"] and "packed" in enc["kernel"], enc
    assert leg["worst_err_over_bound"] <= 1.0


def test_projected_tables_hbm_leg_launch_is_verified(hip_lib):
    """The roofline leg of the instance the default line TIMES (VERDICT r5 #1c): mvin_gather_attn_l2_prj_fwd over the three projected
    tables of a 4 M-row entity table (3.07 GB), checked against the float64 evaluation of the UNPROJECTED formulas on the original table."""
    leg = bench.hbm_leg(torch.device("cuda:0"), 64, 32, "f32", iters=1, warmup=1, prj=True)
    assert leg["form"] == "prj" and "gather_attn_l2_wpp_kernel<32>" in leg["kernel"], leg
    assert leg["outputs_finite"] and leg["verified"] and leg["worst_err_over_bound"] <= 1.0, leg
    assert leg["table_rows"] == bench.HBM_LEG_PRJ_ROWS and leg["table_bytes"] == 3 * bench.HBM_LEG_PRJ_ROWS * 64 * 4
    assert leg["bytes_per_pair"] == (1 + 64 + 1024) * 256 + 33 * 32 * 8 + 256 + 4
    # BASELINE C2's shape: the wave-per-parent kernel over the plain adjacency reads projected tables too
    c2 = bench.hbm_leg(torch.device("cuda:0"), 32, 16, "f32", iters=1, warmup=1, prj=True, n_rows=2_000_000)
    assert c2["verified"] and c2["form"] == "prj", c2


@pytest.mark.parametrize("dtype,rows", [("f32", 9_000_000), ("bf16", 17_500_000)])
def test_table_above_4gib(dtype, rows, hip_lib):
    """D = 128: 512-byte (fp32) / 256-byte (bf16) rows, table > 2^32 bytes -> 64-bit row addressing in the gather waves."""
    dev = torch.device("cuda:0")
    D, K, nR, P = 128, 32, 9, 8192
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    table = (torch.rand((rows, D), device=dev, generator=g) - 0.5).to(dt)
    assert table.numel() * table.element_size() > (1 << 32)
    adj_e = torch.randint(0, rows, (rows, K), device=dev, generator=g, dtype=torch.int32)
    adj_r = torch.randint(0, nR, (rows, K), device=dev, generator=g, dtype=torch.int32)
    # parents in the last rows too: offsets beyond 4 GiB
    parents = torch.randint(0, rows, (P,), device=dev, generator=g, dtype=torch.int32)
    parents[:64] = rows - 1 - torch.arange(64, device=dev, dtype=torch.int32)
    t0 = torch.rand(nR, device=dev, generator=g)
    t1 = torch.rand(nR, device=dev, generator=g)
    W = (torch.rand((3, D, D), device=dev, generator=g) - 0.5) / D ** 0.5
    q = torch.rand((P, D), device=dev, generator=g)
    b = torch.rand((3, D), device=dev, generator=g) - 0.5
    args = (table, adj_e, adj_r, parents, t0, t1, W[0], W[1], b[0], b[1], q, W[2], b[2])
    out = ops.gather_attn_l2(*args, P, 1, K, D, nR)
    torch.cuda.synchronize()
    max_abs, worst, ok = bench.check_l2_launch(out, *args, K, n_check=256)
    assert ok, (max_abs, worst)
    # the first 64 parents (rows at the very end of the table) explicitly
    r0, r1 = bench.l2_reference_f64(table, adj_e, adj_r, parents[:64], t0, t1, W[0], W[1], b[0], b[1], q[:64], W[2], b[2], K)
    for got, ref in ((out[0][:64].double(), r0), (out[1][:64].double(), r1)):
        assert bool(((got - ref).abs() <= 1e-5 * ref.abs() + 1e-6).all())
