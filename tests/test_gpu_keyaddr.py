"""-m gpu: mvin_key_addressing_fwd (per-pair ripple sets) on its own, through the C-ABI, against a float64
restatement of model.py:161-240's attention reads -- for both kernels behind the entry point: the streaming
LDS-DMA pipeline (mvin_keyaddr_stream.hip: P >= 1, Nm <= 64, V block <= 3 KB) and the register-resident one it
falls back to (mvin_keyaddr.hip).  Ragged memory counts, 1-3 hops, h-set on / off, fp32 and bf16 tables, batches
around the kernel's wave count."""
import numpy as np
import pytest
import torch

from mvin_amd import ops

from parity import assert_close

pytestmark = pytest.mark.gpu


def reference(E, RK, w, items, mh, mr, mt, P):
    """o-vectors [B, (w is not None) + P, D] in float64: model.py:189-195 (h-set) and :214-229 (hops)."""
    E, RK = E.astype(np.float64), RK.astype(np.float64)
    v = E[items]                                               # [B, D]
    outs = []
    if w is not None:
        h0 = E[mh[0]]                                          # [B, Nm, D]
        s = h0 @ w.astype(np.float64)                          # user term and bias are constant over m: cancel
        p = np.exp(s - s.max(1, keepdims=True))
        p /= p.sum(1, keepdims=True)
        outs.append((p[..., None] * h0).sum(1))
    for hop in range(P):
        h, t = E[mh[hop]], E[mt[hop]]
        Rh = np.einsum("bmij,bmj->bmi", RK[mr[hop]], h)        # :214-217
        s = np.einsum("bmi,bi->bm", Rh, v)                     # :220
        p = np.exp(s - s.max(1, keepdims=True))
        p /= p.sum(1, keepdims=True)                           # :223
        outs.append((p[..., None] * t).sum(1))                 # :229
    return np.stack(outs, 1)


def run_case(D, Nm, P, nR, B, bf16, with_set, seed, nE=700):
    rng = np.random.default_rng(seed)
    dev = "cuda:0"
    E = (rng.normal(size=(nE, D)) * 0.5).astype(np.float32)
    RK = (rng.normal(size=(nR, D, D)) / np.sqrt(D)).astype(np.float32)
    w = rng.normal(size=D).astype(np.float32) if with_set else None
    items = rng.integers(0, nE, B)
    nh = max(1, P)
    mh = [rng.integers(0, nE, (B, Nm)).astype(np.int32) for _ in range(nh)]
    mr = [rng.integers(0, nR, (B, Nm)).astype(np.int32) for _ in range(P)]
    mt = [rng.integers(0, nE, (B, Nm)).astype(np.int32) for _ in range(P)]
    Et = torch.from_numpy(E).to(dev)
    if bf16:
        Et = Et.to(torch.bfloat16)
        E = Et.float().cpu().numpy()                            # the values the kernel reads
    # V[b, r, :] = E[item_b] . R_KGE[r]  (the reassociation tests/test_oracle_kat.py checks), in float64 -> fp32
    V = np.einsum("bi,rij->brj", E[items].astype(np.float64), RK.astype(np.float64)).astype(np.float32)
    n_o = P + (1 if with_set else 0)
    out = torch.full((B, n_o * D), float("nan"), dtype=torch.float32, device=dev)
    tl = lambda lst: [torch.from_numpy(x).to(dev) for x in lst]
    ops.key_addressing(Et, torch.from_numpy(V).to(dev) if P else None, torch.from_numpy(w).to(dev) if with_set else None,
                       tl(mh), tl(mr), tl(mt), P, out, n_o * D, nR)
    torch.cuda.synchronize()
    ref = reference(E, RK, w, items, mh, mr, mt, P).reshape(B, n_o * D)
    assert_close(out.cpu().numpy(), ref, f"o-vectors D={D} Nm={Nm} P={P} nR={nR} B={B} bf16={bf16} set={with_set}",
                 rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("D", [16, 32, 64, 128])
@pytest.mark.parametrize("Nm", [1, 5, 16, 17, 33, 64])
def test_memory_counts(D, Nm, bf16, hip_lib):
    if bf16 and D == 16:
        pytest.skip("bf16 rows of 32 bytes: not a table shape the library takes")
    nR = 3072 // (4 * D)                        # largest V block the streaming kernel takes
    run_case(D, Nm, 2, nR, 37, bf16, True, seed=D + Nm)


@pytest.mark.parametrize("P,with_set", [(1, True), (1, False), (2, False), (3, True), (0, True)])
@pytest.mark.parametrize("D", [32, 64])
def test_hops_and_h_set(D, P, with_set, hip_lib):
    run_case(D, 64, P, 9, 29, False, with_set, seed=100 + P)


@pytest.mark.parametrize("B", [1, 2, 4095, 4097, 9001])
def test_batches_around_the_wave_count(B, hip_lib):
    """16 single-wave workgroups per CU = 4096 waves: fewer pairs than waves, one more, a ragged multiple."""
    run_case(64, 64, 2, 9, B, False, True, seed=B, nE=5000)


def test_fallback_kernel_large_v_block_and_many_memories(hip_lib):
    """Shapes the streaming kernel does not take go to the register-resident one: nR * D * 4 > 3 KB, Nm > 64."""
    run_case(64, 16, 1, 39, 33, False, True, seed=7)            # amazon-book-like: 39 relations
    run_case(32, 100, 2, 12, 9, False, True, seed=8)


@pytest.mark.parametrize("udtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("shape", [(64, 64, 2, 9), (32, 64, 2, 12), (64, 16, 1, 39), (32, 100, 2, 12), (64, 5, 3, 9)],
                         ids=lambda s: "D%dNm%dP%dR%d" % s)
def test_users_feed_equals_per_pair_arrays(shape, udtype, hip_lib):
    """mvin_key_addressing_users_fwd (pair b reads user_triplet_set[users[b]]) against mvin_key_addressing_fwd fed
    with the arrays train.py:117-120 would assemble: same kernels, same arithmetic -> identical bits."""
    D, Nm, P, nR = shape
    rng = np.random.default_rng(D + Nm + P)
    dev, nE, nU, B = "cuda:0", 900, 37, 301
    E = torch.from_numpy((rng.normal(size=(nE, D)) * 0.5).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.normal(size=D).astype(np.float32)).to(dev)
    V = torch.from_numpy(rng.normal(size=(B, nR, D)).astype(np.float32)).to(dev)
    uts = np.empty((nU, P, 3, Nm), dtype=np.int32)
    uts[:, :, 0] = rng.integers(0, nE, (nU, P, Nm))
    uts[:, :, 1] = rng.integers(0, nR, (nU, P, Nm))
    uts[:, :, 2] = rng.integers(0, nE, (nU, P, Nm))
    users = rng.integers(0, nU, B)
    uts_d, users_d = torch.from_numpy(uts).to(dev), torch.from_numpy(users).to(dev).to(udtype)
    n_o = P + 1
    a = torch.full((B, n_o * D), float("nan"), dtype=torch.float32, device=dev)
    b = torch.full_like(a, float("nan"))
    ops.key_addressing_users(E, V, w, uts_d, users_d, P, a, n_o * D, nR)
    sel = uts_d[users_d.long()]
    mk = lambda x: [sel[:, i, x].contiguous() for i in range(P)]
    ops.key_addressing(E, V, w, mk(0), mk(1), mk(2), P, b, n_o * D, nR)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)
