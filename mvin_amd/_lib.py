"""ctypes binding of libmvin_hip.so (include/mvin_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared
object is missing or fails to load, importing a compute entry point raises immediately.
"""
import ctypes as C
import os

from . import build as _build

MAX_SRC = 8
ABI_VERSION = 12          # include/mvin_hip.h: MVIN_ABI_VERSION

_c_f32p = C.c_void_p
_c_i32p = C.c_void_p


class LinearArgs(C.Structure):
    """mvin_linear_args (include/mvin_hip.h)."""
    _fields_ = [
        ("src", C.c_void_p * MAX_SRC),
        ("ids", C.c_void_p * MAX_SRC),
        ("nsrc", C.c_int),
        ("Dsrc", C.c_int),
        ("Dout", C.c_int),
        ("rows", C.c_int64),
        ("W", C.c_void_p),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p),
        ("rows_per_group", C.c_int),
        ("relu", C.c_int),
        ("ids64", C.c_int),
        ("sum_sources", C.c_int),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("nz", C.c_int),
        ("w_zstride", C.c_int64),
        ("bias_zstride", C.c_int64),
        ("out_zstride", C.c_int64),
        ("score_u", C.c_void_p),
        ("score_out", C.c_void_p),
        ("sigmoid_out", C.c_void_p),
        ("src_bf16", C.c_int),
        ("src_rows", C.c_int64),
    ]


class WgradProblem(C.Structure):
    """mvin_wgrad_problem (include/mvin_hip.h)."""
    _fields_ = [
        ("lin", LinearArgs),
        ("dY", C.c_void_p),
        ("ldy", C.c_int64),
        ("dy_zstride", C.c_int64),
        ("mask", C.c_void_p),
        ("ldm", C.c_int64),
        ("mask_zstride", C.c_int64),
        ("dW", C.c_void_p),
        ("dw_zstride", C.c_int64),
        ("db", C.c_void_p),
        ("db_zstride", C.c_int64),
    ]


class ScoreL2Args(C.Structure):
    """mvin_score_l2_args (include/mvin_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "entity_emb", "adj_entity", "adj_relation", "relation_kge", "h_set_w", "user_mlp_W", "user_mlp_b", "t0", "t1",
        "W0", "b0", "W1", "b1", "W2", "b2", "A0", "a0", "A1", "a1", "Wmix", "bmix", "items", "mem_h", "mem_r", "mem_t",
        "uts", "users", "V", "o_cat", "parents", "nagg0", "nagg1", "user_o", "item_emb", "scores", "sig")] + [
        ("B", C.c_int64)] + [(n, C.c_int) for n in ("D", "K", "P", "Nm", "n_entity", "n_relation", "table_bf16", "n_user")] + [
        ("enc_entity", C.c_void_p), ("enc_relation", C.c_void_p), ("group_ws", C.c_void_p),
        ("user_records", C.c_void_p), ("depth", C.c_int), ("prj_tables", C.c_void_p), ("ka_er", C.c_void_p),
        ("ka_flash", C.c_void_p), ("item_order_ws", C.c_void_p), ("agg_tables", C.c_void_p), ("fold_ws", C.c_void_p), ("fold_gather", C.c_int)]


# name -> (restype, argtypes); mirrors include/mvin_hip.h one to one.
SIGNATURES = {
    "mvin_abi_version": (C.c_int, []),
    "mvin_last_error": (C.c_char_p, []),
    "mvin_debug_read_trace": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mvin_l2_tail_fwd": (C.c_int, [C.c_void_p] * 15 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]),
    "mvin_l2_tail_supported": (C.c_int, [C.c_int]),
    "mvin_score_l2_fwd": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mvin_score_small_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mvin_mix_neighbor_vectors_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3),
    "mvin_score_small_supported": (C.c_int, [C.c_int] * 5),
    "mvin_project_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int] + [C.c_void_p] * 6),
    "mvin_project_tables_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "mvin_project_tables": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p] * 2),
    "mvin_gather_attn_l2_prj_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p] * 3),
    "mvin_order_by_key_ws_elems": (C.c_size_t, [C.c_int64]),
    "mvin_order_by_key": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvin_gather_attn_l2_prj_ordered_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p] * 3),
    "mvin_gather_attn_l2_agg_supported": (C.c_int, [C.c_int] * 4),
    "mvin_entity_aggregates_elems": (C.c_size_t, [C.c_int] * 2),
    "mvin_entity_aggregates": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 2),
    "mvin_gather_attn_l2_agg_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p] * 3),
    "mvin_score_l2_folded_supported": (C.c_int, [C.c_int] * 4),
    "mvin_fold_tables_elems": (C.c_size_t, [C.c_int] * 2),
    "mvin_fold_tables": (C.c_int, [C.c_void_p] * 15 + [C.c_int] * 4 + [C.c_void_p] * 2),
    "mvin_fold_tables_ex": (C.c_int, [C.c_void_p] * 15 + [C.c_int] * 5 + [C.c_void_p] * 2),
    "mvin_score_l2_folded_gather_supported": (C.c_int, [C.c_int] * 4),
    "mvin_score_l2_folded_gather_fwd": (C.c_int, [C.c_void_p] * 13 + [C.c_int64] + [C.c_int] * 4 + [C.c_void_p] * 4),
    "mvin_score_l2_folded_fwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int64] + [C.c_int] * 4 + [C.c_void_p] * 6),
    "mvin_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mvin_scatter_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mvin_linear_wgrad_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mvin_shard_space_ids": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvin_key_addressing_grouped_fwd": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 8 + [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "mvin_key_addressing_grouped_rec_fwd": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 8 + [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "mvin_project_relations_elems": (C.c_size_t, [C.c_int] * 3),
    "mvin_project_relations": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2),
    "mvin_key_addressing_grouped_er_supported": (C.c_int, [C.c_int] * 6),
    "mvin_key_addressing_flash_supported": (C.c_int, [C.c_int] * 5),
    "mvin_key_addressing_flash_tables_elems": (C.c_size_t, [C.c_int] * 5),
    "mvin_key_addressing_flash_prepare": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 2),
    "mvin_key_addressing_flash_ws_elems": (C.c_size_t, [C.c_int64, C.c_int]),
    "mvin_key_addressing_flash_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int64] + [C.c_int] * 7 + [C.c_void_p] * 4),
    "mvin_key_addressing_grouped_er_fwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int] * 8 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mvin_user_records_len": (C.c_int, [C.c_int] * 3),
    "mvin_user_records_supported": (C.c_int, [C.c_int] * 5),
    "mvin_build_user_records": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvin_key_addressing_grouped_supported": (C.c_int, [C.c_int] * 4),
    "mvin_ent_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mvin_rel_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mvin_expand_ids": (C.c_int, [_c_i32p, _c_i32p, C.c_void_p, _c_i32p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, _c_i32p, _c_i32p, C.c_void_p]),
    "mvin_rel_score": (C.c_int, [_c_f32p, _c_f32p, C.c_int, C.c_int, _c_f32p, C.c_void_p]),
    "mvin_linear_fwd": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    "mvin_gather_attn_fwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p,
                                       _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_gather_attn_l2_fwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p,
                                          _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, _c_f32p, _c_f32p,
                                          _c_f32p, C.c_int, C.c_void_p]),
    "mvin_gather_attn_l2_fwd_i64": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, C.c_void_p, _c_f32p, _c_f32p, _c_f32p,
                                              _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, _c_f32p, _c_f32p,
                                              _c_f32p, C.c_int, C.c_void_p]),
    "mvin_gather_attn_l2_supported": (C.c_int, [C.c_int, C.c_int]),
    "mvin_gather_attn_l2_enc_fwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, C.c_void_p, C.c_int, _c_f32p, _c_f32p, _c_f32p,
                                              _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, _c_f32p, C.c_int, C.c_void_p]),
    "mvin_gather_attn_l2_enc_supported": (C.c_int, [C.c_int, C.c_int]),
    "mvin_gather_attn_l2_prj_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvin_encode_adjacency": (C.c_int, [_c_i32p, _c_i32p, C.c_int, C.c_int, _c_i32p, _c_i32p, _c_i32p, C.c_void_p]),
    "mvin_gather_attn_l2_variant": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]),
    "mvin_gather_attn_l2_variant_ex": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "mvin_probe_gather_l2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "mvin_agg_fwd": (C.c_int, [_c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int,
                               C.c_int, C.c_int, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_key_addressing_fwd": (C.c_int, [_c_f32p, _c_f32p, _c_f32p, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, C.c_int64, C.c_int,
                                          C.c_void_p]),
    "mvin_key_addressing_supported": (C.c_int, [C.c_int, C.c_int]),
    "mvin_count_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mvin_group_pairs_by_user": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvin_key_addressing_users_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int64, C.c_int, C.c_void_p]),
    "mvin_row_softmax_fwd": (C.c_int, [_c_f32p, C.c_int64, C.c_int, _c_f32p, C.c_void_p]),
    "mvin_gather_mix_fwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, C.c_int64, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, C.c_int, C.c_void_p]),
    "mvin_sample_adjacency": (C.c_int, [C.c_void_p, _c_i32p, _c_i32p, C.c_int, C.c_int, C.c_uint64, _c_i32p,
                                        _c_i32p, C.c_void_p]),
    "mvin_build_ripple_sets": (C.c_int, [C.c_void_p, _c_i32p, _c_i32p, C.c_void_p, _c_i32p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_uint64, _c_i32p, C.c_void_p]),
    "mvin_gather_attn_fwd_ex": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p,
                                          _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_void_p]),
    "mvin_agg_fwd_ex": (C.c_int, [_c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int,
                                  C.c_int, C.c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_l2_adam_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mvin_l2_adam_multi_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_int,
                                         _c_f32p, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mvin_eltwise": (C.c_int, [C.c_int, C.c_int64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "mvin_scatter_add_rows": (C.c_int, [_c_f32p, C.c_void_p, C.c_int, _c_f32p, C.c_int64, C.c_int, C.c_float,
                                        C.c_void_p]),
    "mvin_linear_wgrad": (C.c_int, [C.POINTER(LinearArgs), _c_f32p, C.c_int64, C.c_int64, _c_f32p, C.c_int64,
                                    C.c_int64, _c_f32p, C.c_int64, _c_f32p, C.c_int64, C.c_void_p]),
    "mvin_agg_bwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p, C.c_int64,
                               C.c_int, C.c_int, C.c_int, _c_f32p, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_rel_score_bwd": (C.c_int, [_c_f32p, _c_f32p, _c_f32p, C.c_int, C.c_int, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_key_addressing_bwd": (C.c_int, [_c_f32p, _c_f32p, _c_f32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          _c_f32p, C.c_int64, C.c_float, _c_f32p, _c_f32p, _c_f32p, C.c_void_p]),
    "mvin_key_addressing_bwd_reg": (C.c_int, [_c_f32p, _c_f32p, _c_f32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              _c_f32p, C.c_int64, C.c_float, _c_f32p, _c_f32p, _c_f32p, C.c_int,
                                              _c_f32p, _c_f32p, C.c_void_p, C.c_int, C.c_void_p]),
    "mvin_key_addressing_bwd_adds_item_grad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvin_ripple_attn_fwd": (C.c_int, [_c_f32p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, C.c_int64,
                                       C.c_void_p]),
    "mvin_ripple_attn_fwd_ex": (C.c_int, [C.c_void_p, _c_i32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _c_f32p, C.c_int64, C.c_int,
                                          C.c_void_p]),
}

_lib = None


class MvinHipError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load():
    """Load (once) and return the ctypes handle; raise loudly when the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MvinHipError(
            f"{path} is missing: build it with `python -m mvin_amd.build` (hipcc, gfx950). "
            "mvin_amd has no CPU fallback.")
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
    # It must be in the process BEFORE this library is opened so that our DT_NEEDED
    # libamdhip64.so.7 resolves to the same runtime instance (streams and device pointers
    # are shared with torch); otherwise /opt/rocm's copy is loaded as a second runtime.
    import torch  # noqa: F401
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.mvin_abi_version()
    if ver != ABI_VERSION:
        raise MvinHipError(f"libmvin_hip.so ABI version {ver}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mvin_last_error().decode(errors="replace")
        raise MvinHipError(f"{what or 'mvin call'} failed (rc={rc}): {msg}")
