"""CPU: libmvin_hip.so builds, loads and exports every symbol include/mvin_hip.h declares;
argument validation returns error codes before anything touches a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mvin_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvin_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for name in ("mvin_expand_ids", "mvin_gather_attn_fwd", "mvin_agg_fwd", "mvin_linear_fwd",
                 "mvin_ripple_attn_fwd", "mvin_rel_score", "mvin_last_error", "mvin_abi_version"):
        assert name in fns


def test_library_exports_every_declared_symbol(hip_lib):
    from mvin_amd import _lib
    for name in header_functions():
        assert hasattr(hip_lib, name), f"{name} declared in include/mvin_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in mvin_amd/_lib.py"
    assert hip_lib.mvin_abi_version() == _lib.ABI_VERSION == 12


def test_dynamic_symbol_table_is_exactly_the_header(hip_lib):
    """The library is built with -fvisibility=hidden and linked with csrc/libmvin_hip.map: what `nm -D` lists as
    defined is the header's declarations, nothing internal (no mangled mvin:: symbols, no kernel host stubs)."""
    import subprocess
    from mvin_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == header_functions()


def test_single_hip_runtime_in_process(hip_lib):
    with open("/proc/self/maps") as f:
        rts = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    assert len(rts) == 1, rts


def test_id_buffer_sizes(hip_lib):
    # entities: B*(1+K+K^2), relations: B*(K+K^2)
    assert hip_lib.mvin_ent_elems(3, 4, 2) == 3 * (1 + 4 + 16)
    assert hip_lib.mvin_rel_elems(3, 4, 2) == 3 * (4 + 16)
    assert hip_lib.mvin_rel_elems(3, 4, 0) == 0


def test_argument_errors_are_reported_not_thrown(hip_lib):
    from mvin_amd import _lib
    rc = hip_lib.mvin_rel_score(None, None, 4, 8, None, None)
    assert rc == -1 and b"null" in hip_lib.mvin_last_error()
    rc = hip_lib.mvin_expand_ids(None, None, None, None, 4, 3, 1, 10, None, None, None)
    assert rc < 0
    a = _lib.LinearArgs()
    a.nsrc, a.Dsrc, a.Dout = 1, 6, 8  # Dsrc not a multiple of 4
    assert hip_lib.mvin_linear_fwd(C.byref(a), None) == -2
    assert b"Dsrc" in hip_lib.mvin_last_error()
    a.nsrc, a.Dsrc, a.Dout = 9, 8, 8
    assert hip_lib.mvin_linear_fwd(C.byref(a), None) == -2
    # D not a multiple of 4 / out of range
    one = C.c_void_p(16)
    rc = hip_lib.mvin_agg_fwd(one, one, None, None, one, None, 2, 2, 2, 6, one, None, None)
    assert rc == -2 and b"D=6" in hip_lib.mvin_last_error()
    rc = hip_lib.mvin_ripple_attn_fwd(one, one, None, one, None, None, 0, 1, 4, 8, 3, one, 8, None)
    assert rc == -1  # mode 0 without V / rel_ids
    with pytest.raises(_lib.MvinHipError):
        _lib.check(rc, "mvin_ripple_attn_fwd")


def test_round6_entry_points_validate_before_launching(hip_lib):
    """The aggregates / folded-tail entry points (ABI 12): sizes and shape support are plain host arithmetic, null pointers and
    unsupported shapes come back as error codes before anything is launched."""
    one = C.c_void_p(16)
    nE, nR = 1000, 9
    assert hip_lib.mvin_entity_aggregates_elems(nE, 64) == 2 * nE * 64
    assert hip_lib.mvin_fold_tables_elems(nE, 64) == 6 * nE * 64 + 12 * 64 * 64 + 3 * 64
    assert hip_lib.mvin_order_by_key_ws_elems(1000) > 1000
    # which shapes the forms take: aggregates dim 64 only; folded tail dim 64 and dim 32; the gather form dim 64, K <= 32
    assert hip_lib.mvin_gather_attn_l2_agg_supported(64, 32, nE, nR) == 1 and hip_lib.mvin_gather_attn_l2_agg_supported(64, 64, nE, nR) == 1
    assert hip_lib.mvin_gather_attn_l2_agg_supported(32, 16, nE, nR) == 0 and hip_lib.mvin_gather_attn_l2_agg_supported(64, 128, nE, nR) == 0
    assert hip_lib.mvin_score_l2_folded_supported(32, 16, nE, nR) == 1 and hip_lib.mvin_score_l2_folded_supported(32, 64, nE, nR) == 0
    assert hip_lib.mvin_score_l2_folded_supported(128, 32, nE, nR) == 0 and hip_lib.mvin_score_l2_folded_supported(64, 32, 1 << 25, nR) == 0
    assert hip_lib.mvin_score_l2_folded_gather_supported(64, 32, nE, nR) == 1 and hip_lib.mvin_score_l2_folded_gather_supported(64, 64, nE, nR) == 0
    assert hip_lib.mvin_score_l2_folded_supported(64, 32, nE, 5000) == 0          # more relation logits than the kernels keep in LDS
    # null pointers
    assert hip_lib.mvin_entity_aggregates(None, one, one, None, 32, 64, nE, nR, one, None) == -1
    assert hip_lib.mvin_fold_tables(one, one, one, None, None, None, one, None, one, None, one, None, one, None, one, 32, 64, nE, nR, one, None) == -1
    assert hip_lib.mvin_score_l2_folded_fwd(None, one, one, one, None, None, None, one, one, one, None, one, 8, 32, 64, nE, nR, None, None, None,
                                            one, None, None) == -1
    # both / neither id width
    assert hip_lib.mvin_score_l2_folded_fwd(one, one, one, one, one, None, None, one, one, one, None, one, 8, 32, 64, nE, nR, None, None, None,
                                            one, None, None) == -1
    assert hip_lib.mvin_order_by_key(None, None, 8, one, one, None) == -1
    # a shape no kernel takes: -3, with the reason in the error string
    rc = hip_lib.mvin_fold_tables(one, one, one, None, one, None, one, None, one, None, one, None, one, None, one, 128, 64, nE, nR, one, None)
    assert rc == -3 and b"K in {16, 32, 64}" in hip_lib.mvin_last_error()
    rc = hip_lib.mvin_score_l2_folded_gather_fwd(one, one, one, one, None, None, None, None, one, one, one, None, one, 8, 64, 64, nE, nR, None,
                                                 one, None, None)
    assert rc == -3


def test_ops_refuse_cpu_tensors(hip_lib):
    import torch
    from mvin_amd import _lib, ops
    with pytest.raises(_lib.MvinHipError):
        ops.rel_score(torch.zeros(3, 8), torch.zeros(24, 1))


def test_header_is_plain_c_and_cpp():
    """include/mvin_hip.h is the whole boundary: it must compile as C99 and as C++11 on its own."""
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mvin_hip.h")
    for cc, lang, std in (("gcc", "c", "-std=c99"), ("g++", "c++", "-std=c++11")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        r = subprocess.run([cc, "-fsyntax-only", "-x", lang, std, "-Wall", "-Werror", hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
