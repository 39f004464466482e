"""Shared helpers of the parity tests: run the HIP path and both oracles on one case."""
import numpy as np

from mvin_amd.params import init_params

# |delta| <= RTOL*|ref| + ATOL against the fp32 mirror (BASELINE.json north_star: 1e-5
# relative; the absolute floor is the fp32 round-off of O(1) sums)
RTOL = 1e-5
ATOL = 1e-6


def run_oracles(args, case, params):
    from oracle import equations_fp64, mirror_fp32
    m = mirror_fp32.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                            case.memories_h, case.memories_r, case.memories_t)
    e = equations_fp64.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                               case.memories_h, case.memories_r, case.memories_t)
    return m, e


def run_hip(args, case, params, want_probs=True, fused=None, hoist=False, table_dtype="f32"):
    import torch
    from mvin_amd.model import MVIN
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0", fused=fused, hoist=hoist, table_dtype=table_dtype)
    dev = model.device
    out = model.forward_device(
        torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
        [torch.from_numpy(m).to(dev) for m in case.memories_h],
        [torch.from_numpy(m).to(dev) for m in case.memories_r],
        [torch.from_numpy(m).to(dev) for m in case.memories_t], want_probs=want_probs)
    torch.cuda.synchronize()
    return model, out


def assert_close(got, ref, what, rtol=RTOL, atol=ATOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    err = np.abs(got - ref)
    bound = rtol * np.abs(ref) + atol
    bad = err > bound
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} outside tolerance; max abs err "
                           f"{err.max():.3e}, worst ratio {(err / bound).max():.2f}")


def check_case(args, case, params=None, seed=0, fused=None, hoist=False):
    """HIP path vs fp32 mirror (tolerance above) and vs fp64 (error no worse than 4x the
    mirror's own fp32 round-off, with a 1e-6 floor)."""
    if params is None:
        params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=seed,
                             random_agg_bias=True)
    m, e = run_oracles(args, case, params)
    _, out = run_hip(args, case, params, fused=fused, hoist=hoist, want_probs=not hoist)
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), "scores vs fp32 mirror")
    assert_close(out.scores_normalized.cpu().numpy(), m.scores_normalized.numpy(), "sigmoid scores")
    assert_close(out.user_o.cpu().numpy(), m.user_o.numpy(), "user_o")
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), "item_embeddings")
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"
    if hoist:   # attention outputs come from the faithful path only (requested with want_probs)
        return out, m, e
    for h, (pg, pm) in enumerate(zip(out.importance_list, m.importance_list)):
        if pm is None:
            assert pg is None
        else:
            assert_close(pg.cpu().numpy(), pm.numpy(), f"importance_list[{h}]")
    return out, m, e
