"""Run configuration: the ``args`` namespace the reference's MVIN reads.

Mirrors the flags of src/model/MVIN/parser.py:8-57 (defaults kept) and the
``--ablation`` presets of src/model/MVIN/parameter_ablation.py:3-179 (ints are
turned into bools exactly as :167-175 does).  Only the fields MVIN._parse_args
(model.py:17-47) and the model body read matter to the hot path; the rest are
carried so a reference ``args`` object can be passed through unchanged.
"""
from types import SimpleNamespace

# parser.py:8-57 defaults
_DEFAULTS = dict(
    dataset="music", aggregator="sum", n_epochs=20, neighbor_sample_size=8, p_hop=1,
    user_agg_hop=0, n_memory=16, dim=8, h_hop=3, batch_size=512, l2_weight=1e-4,
    l2_agg_weight=1e-6, kge_weight=1e-2, lr=5e-4, tolerance=2, early_decrease_lr=2,
    early_stop=3, update_item_emb="transform_matrix", h0_att="st_att_h_set",
    model_select="KGCN", n_mix_hop=2, load_pretrain_emb=False, save_default_model=False,
    save_final_model=True, save_record_user_list=False, show_topk_mode=False,
    use_neighbor_rate=0, save_model_name="model1", new_load_data=False, log_name="",
    SW_stage=0, top_k=0, ablation="all", abla_exp=0, SW=1, User_orient=1,
    User_orient_rela=1, User_orient_kg_eh=1, PS_W_ft=1, PS_O_ft=1, wide_deep=1,
    PS_only=0, HO_only=0, attention_cast_st=0, path=None,
)

# parameter_ablation.py:4-165 -- name -> (SW, User_orient, User_orient_rela,
# User_orient_kg_eh, PS_O_ft, wide_deep, PS_only, HO_only)
ABLATIONS = {
    "all":                       (1, 1, 1, 1, 1, 1, 0, 0),
    "no_sw":                     (0, 1, 1, 1, 1, 1, 0, 0),
    "no_kg_eh_uo":               (1, 1, 1, 0, 1, 1, 0, 0),
    "no_kg_eh_uo_sw":            (1, 1, 1, 0, 1, 1, 0, 0),
    "no_uo_and_no_kg_eh_uo":     (1, 0, 1, 0, 1, 1, 0, 0),
    "no_uo_and_no_kg_eh_uo_sw":  (1, 0, 1, 0, 1, 1, 0, 0),
    "no_uor_and_no_kg_eh_uo":    (1, 1, 0, 0, 1, 1, 0, 0),
    "no_uor_and_no_kg_eh_uo_sw": (1, 1, 0, 0, 1, 1, 0, 0),
    "no_uo":                     (1, 0, 1, 1, 1, 1, 0, 0),
    "no_uor":                    (1, 1, 0, 1, 1, 1, 0, 0),
    "no_wd":                     (1, 1, 1, 1, 1, 0, 0, 0),
    "no_ps_o_ft":                (1, 1, 1, 1, 0, 1, 0, 0),
    "ps_only":                   (1, 1, 1, 1, 1, 1, 1, 0),
    "ho_only":                   (1, 1, 1, 0, 1, 1, 0, 1),
    "ho_only_uo_kg_eh":          (1, 1, 1, 1, 1, 1, 0, 1),
    "no_wd_ho_only":             (1, 1, 1, 0, 1, 0, 0, 1),
    "no_uo_ho_only":             (1, 0, 1, 0, 1, 1, 0, 1),
    "no_uor_ho_only":            (1, 1, 0, 0, 1, 1, 0, 1),
}
_SWITCHES = ("SW", "User_orient", "User_orient_rela", "User_orient_kg_eh", "PS_O_ft",
             "wide_deep", "PS_only", "HO_only")


def parameter_env(args):
    """parameter_ablation.py:3-179: apply the preset named by ``args.ablation`` (if it
    is one of the known names; otherwise the individual switches are left as given),
    then coerce the int switches to bools (:167-175)."""
    preset = ABLATIONS.get(getattr(args, "ablation", None))
    if preset is not None:
        for name, val in zip(_SWITCHES, preset):
            setattr(args, name, val)
    for name in ("abla_exp",) + _SWITCHES:
        setattr(args, name, int(getattr(args, name)) == 1)
    return args


def make_args(**overrides):
    """Build an ``args`` namespace with parser.py defaults, then apply overrides.
    Switches passed explicitly win over the ablation preset."""
    d = dict(_DEFAULTS)
    explicit = {k: v for k, v in overrides.items() if k in _SWITCHES}
    d.update(overrides)
    args = parameter_env(SimpleNamespace(**d))
    for k, v in explicit.items():
        setattr(args, k, bool(v))
    args.top_k = (args.top_k == 1)
    args.attention_cast_st = (args.attention_cast_st == 1)
    return args


def tree_depth(args):
    """L = n_mix_hop * h_hop (model.py:250)."""
    return args.n_mix_hop * args.h_hop
