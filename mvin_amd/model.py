"""MVIN model -- the reference's call surface (src/model/MVIN/model.py:6-444) on top of
libmvin_hip.so.  Host code in Python on PyTorch-ROCm (device memory + stream only); every
op of the scoring path is a hand-written gfx950 kernel reached through the C ABI.

    model = MVIN(args, n_user, n_entity, n_relation, adj_entity, adj_relation)
    model.get_scores(sess, feed_dict) -> (item_indices, sigmoid_scores)     model.py:443-444
    model.eval(sess, feed_dict)       -> (auc, acc, f1)                      model.py:419-426
    model.eval_case_study(sess, feed) -> 7-tuple                             model.py:428-441
    model.train(sess, feed_dict)      -> (None, loss)                        model.py:416-417

``sess`` is accepted and ignored (there is no TF session); feed keys are the placeholder
sentinels ``model.user_indices / item_indices / labels / memories_{h,r,t}[i]``
(train.py:113-120, util.py:209-230).  Unlike the reference graph, whose static reshapes
bake ``batch_size`` in (model.py:251,281,296), any batch length is accepted.

How the forward is arranged (see DESIGN.md for the data layout and the kernel roofline):
  * ids of levels 0..L-1 are expanded on device (mvin_expand_ids); level L is never
    materialised -- the deepest aggregator hop reads the adjacency rows itself and gathers
    the K^L entity rows straight into a wave-level softmax / weighted sum
    (mvin_gather_attn_fwd);
  * the user-oriented projection (model.py:270-283) is applied AFTER the weighted sum on
    the deepest level ((sum_k p_k E[y_k]).W + c == sum_k p_k (E[y_k].W + c) as sum_k p_k = 1)
    and per row on the small upper levels (mvin_linear_fwd with a gathered source);
  * attention logits use the nR-entry table t[r] = Rel[r].urh_w[D:2D]: the user and self
    terms of aggregators.py:130-133 are constant over k and cancel in the softmax;
  * key addressing (model.py:161-240) uses (R h).v == h.(v R): V[b,r,:] = E[item_b].R_KGE[r]
    is computed once per pair, then each memory needs one D-long dot product.
"""
import os
import weakref
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .aggregators import SumAggregator_urh_matrix
from .params import aggregator_keys, init_params


class Placeholder(object):
    """Hashable stand-in for a tf.placeholder used as a feed_dict key (model.py:49-64)."""

    def __init__(self, name, dtype, shape):
        self.name, self.dtype, self.shape = name, dtype, shape

    def __repr__(self):
        return f"<Placeholder {self.name} {self.dtype} {self.shape}>"


class _SmallOut(object):
    """Results of the single-launch pass: one device buffer [user_o | item_embeddings | scores | sigmoid scores]; the four
    tensors are views created on access (same attribute names as the namespace the other schedules return)."""
    __slots__ = ("_buf", "_B", "_D", "importance_list")

    def __init__(self, buf, B, D):
        self._buf, self._B, self._D, self.importance_list = buf, B, D, []

    @property
    def user_o(self):
        return self._buf[:self._B * self._D].view(self._B, self._D)

    @property
    def item_embeddings(self):
        n = self._B * self._D
        return self._buf[n:2 * n].view(self._B, self._D)

    @property
    def scores(self):
        n = 2 * self._B * self._D
        return self._buf[n:n + self._B]

    @property
    def scores_normalized(self):
        n = 2 * self._B * self._D + self._B
        return self._buf[n:n + self._B]


def _lib_flash_elems(m):
    from . import _lib
    return _lib.load().mvin_key_addressing_flash_tables_elems(m.n_entity, m.n_relation, m.dim, m.p_hop, 1 if m.args.PS_O_ft else 0)


class MVIN(object):
    def __init__(self, args, n_user, n_entity, n_relation, adj_entity, adj_relation,
                 params=None, device=None, seed=0, fused=None, table_dtype="f32", hoist=False):
        """``table_dtype="bf16"`` keeps the entity embedding table in bf16 (BASELINE config C5):
        rows are widened to fp32 inside the kernels, all arithmetic stays fp32; scoring only.
        ``hoist``: entity-table mode of the two deepest levels (see ``hoist_entity_tables``):
        False = off (the faithful per-pair gather), True = per-entity tables cached until a
        parameter / the adjacency changes, "step" = rebuilt inside every forward."""
        self.device = torch.device(device or "cuda")
        if table_dtype not in ("f32", "bf16"):
            raise ValueError("table_dtype must be 'f32' or 'bf16'")
        self.table_dtype = table_dtype
        # fused=False (or MVIN_FUSED=0) forces the per-level kernels (used by the tests to
        # check both HIP paths against the oracle)
        self.fused = (os.environ.get("MVIN_FUSED", "1") != "0") if fused is None else bool(fused)
        if hoist not in (False, True, "step"):
            raise ValueError("hoist must be False, True or 'step'")
        self.hoist = hoist
        self.dedup = None            # None: MVIN_L2_ENC / auto; True / False: force the encoded / the plain adjacency
        self._hoisted = None
        self._table_token = None     # see _hoist_key
        self._parse_args(args, adj_entity, adj_relation)
        self._build_inputs()
        self._build_model(n_user, n_entity, n_relation, params, seed)
        self._build_train()
        self._built = True
        self.prepare()

    # ------------------------------------------------------------------ construction
    def _parse_args(self, args, adj_entity, adj_relation):
        """model.py:17-47."""
        self.dataset = getattr(args, "dataset", None)
        self.load_pretrain_emb = getattr(args, "load_pretrain_emb", False)
        self.h_hop = args.h_hop
        self.batch_size = args.batch_size
        self.n_neighbor = args.neighbor_sample_size
        self.p_hop = args.p_hop
        self.dim = args.dim
        self.l2_weight = getattr(args, "l2_weight", 0.0)
        self.l2_agg_weight = getattr(args, "l2_agg_weight", 0.0)
        self.kge_weight = getattr(args, "kge_weight", 0.0)
        self.lr = getattr(args, "lr", 0.0)
        self.save_model_name = getattr(args, "save_model_name", "model1")
        self.n_mix_hop = args.n_mix_hop
        self.n_memory = args.n_memory
        self.update_item_emb = getattr(args, "update_item_emb", None)
        self.h0_att = getattr(args, "h0_att", None)
        self.path = getattr(args, "path", None)
        self.User_orient_rela = bool(args.User_orient_rela)
        self.args = args
        self.aggregator_class = SumAggregator_urh_matrix
        self.agg_fun = self.aggregate_delta_whole if args.wide_deep else self.aggregate
        if self.dim % 4 != 0 or not (4 <= self.dim <= 256):
            raise ValueError("dim must be a multiple of 4 in [4, 256] (16-byte row loads)")
        if tuple(adj_entity.shape) != tuple(adj_relation.shape) or adj_entity.shape[1] != self.n_neighbor:
            raise ValueError("adj_entity/adj_relation must both be [n_entity, neighbor_sample_size]")
        self.set_adjacency(adj_entity, adj_relation)

    def set_adjacency(self, adj_entity, adj_relation):
        """Install a (re-)sampled adjacency: numpy int64 arrays as the reference builds them
        (data_loader_user_set.py:377-378) or device tensors from mvin_amd.data_prep.construct_adj.
        Kept as int32 on the device: halves the id traffic (n_entity < 2^31)."""
        def conv(a):
            if isinstance(a, torch.Tensor):
                return a.to(self.device).to(torch.int32).contiguous()
            return torch.from_numpy(np.asarray(a).astype(np.int32)).to(self.device).contiguous()
        self.adj_entity, self.adj_relation = conv(adj_entity), conv(adj_relation)
        self._hoisted = None
        self._enc = None             # duplicate-slot encoding, built on first use (encoded_adjacency)
        self._generation = getattr(self, "_generation", 0) + 1
        if getattr(self, "_built", False):
            self.prepare()           # the encoding of the NEW adjacency, eagerly (never inside a later forward / capture)

    # a sampled adjacency repeats slots whenever deg < K (data_loader_user_set.py:383-384); the packed-tile fused kernel
    # walks the distinct slots only.  It pays when rows repeat: below this mean fraction of distinct slots per row it is
    # taken ("auto"); MVIN_L2_ENC=0 / 1 (or MVIN.dedup = False / True) force the plain / the encoded path.
    ENC_AUTO_MAX_DISTINCT_FRACTION = 0.75

    def encoded_adjacency(self):
        """(enc_entity, enc_relation, cnt, mean distinct fraction) of the current adjacency (mvin_encode_adjacency),
        built once per set_adjacency; None when the shape has no packed-tile kernel."""
        if self._enc is None:
            if torch.cuda.is_current_stream_capturing():
                # two launches, an allocation and a host sync (.item()): none of that belongs in a captured graph
                raise RuntimeError("the duplicate-slot encoding of the adjacency is not built yet: call model.prepare() "
                                   "(or run one eager forward) before capturing a graph")
            K = self.n_neighbor
            if (K > 128 or int(self.n_relation) > 4096 or int(self.n_entity) > (1 << 24)
                    or not ops.encode_adjacency_supported(self.dim, K)):
                self._enc = False
            else:
                enc_e, enc_r, cnt = ops.encode_adjacency(self.adj_entity, self.adj_relation)
                frac = float(cnt.float().mean().item()) / K       # one host sync per adjacency
                self._enc = (enc_e, enc_r, cnt, frac)
        return self._enc or None

    ENC_AUTO_MIN_PARENTS = 2048      # below: a few tiles per workgroup, the pipeline's fill / drain dominates (B = 512: 68 vs 58 us)

    def _enc_for_l2(self, want_probs=False, n_parents=None):
        """The encoded adjacency when the two deepest levels should take the packed-tile kernel for this call."""
        mode = os.environ.get("MVIN_L2_ENC", "auto") if self.dedup is None else ("1" if self.dedup else "0")
        if mode == "0" or want_probs or self.fused is False:
            return None
        if mode != "1":
            # measured (scripts/ab_enc.sh): the wave-per-parent kernel keeps D = 32, K <= 16 (BASELINE C2: 1.53 vs 1.88 ms)
            # ... unless the call takes the folded tail over per-entity aggregates, which reads the encoding (C2 at bench size)
            if self.dim == 32 and self.n_neighbor <= 16 and not (
                    n_parents is not None and self.n_mix_hop * self.h_hop == 2 and self.args.User_orient and self.fold is not False
                    and self.agg is not False and self.prj is not False and self.entity_emb_matrix.dtype == torch.float32
                    and n_parents * self.n_neighbor >= 10 * self.n_entity and self._fold_shape_ok()):
                return None
            if n_parents is not None and n_parents < self.ENC_AUTO_MIN_PARENTS:
                return None
        if self.entity_emb_matrix.numel() * self.entity_emb_matrix.element_size() >= (1 << 32):
            return None
        # the packed-tile kernel addresses the adjacency and its output rows with 32-bit offsets (fused_packed_applies)
        if self.adj_entity.numel() * 4 >= (1 << 31) or (n_parents is not None and n_parents * self.dim * 4 >= (1 << 31)):
            return None
        enc = self.encoded_adjacency()
        if enc is None or (mode != "1" and enc[3] > self.ENC_AUTO_MAX_DISTINCT_FRACTION):
            return None
        return enc

    def _build_inputs(self):
        """model.py:49-64."""
        self.user_indices = Placeholder("user_indices", "int64", [None])
        self.item_indices = Placeholder("item_indices", "int64", [None])
        self.labels = Placeholder("labels", "float32", [None])
        self.memories_h, self.memories_r, self.memories_t = [], [], []
        for hop in range(max(1, self.p_hop)):
            self.memories_h.append(Placeholder(f"memories_h_{hop}", "int32", [None, self.n_memory]))
            self.memories_r.append(Placeholder(f"memories_r_{hop}", "int32", [None, self.n_memory]))
            self.memories_t.append(Placeholder(f"memories_t_{hop}", "int32", [None, self.n_memory]))

    def _build_model(self, n_user, n_entity, n_relation, params, seed):
        """model.py:69-122 (parameters).  The graph body (:125-159) runs in ``forward_device``."""
        a = self.args
        self.n_user, self.n_entity, self.n_relation = n_user, n_entity, n_relation
        if params is None:
            params = init_params(a, n_user, n_entity, n_relation, seed=seed)
        D, H, M = self.dim, self.h_hop, self.n_mix_hop
        L = M * H

        def dev(x):
            return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device).contiguous()

        self.user_emb_matrix = dev(params["user_emb_matrix"])
        self.entity_emb_matrix = dev(params["entity_emb_matrix"])
        if self.table_dtype == "bf16":
            self.entity_emb_matrix = self.entity_emb_matrix.to(torch.bfloat16).contiguous()
        self.relation_emb_matrix = dev(params["relation_emb_matrix"])
        self.relation_emb_KGE_matrix = dev(params["relation_emb_KGE_matrix"])
        for t, shape in ((self.user_emb_matrix, (n_user, D)), (self.entity_emb_matrix, (n_entity, D)),
                         (self.relation_emb_matrix, (n_relation, D)),
                         (self.relation_emb_KGE_matrix, (n_relation, D, D))):
            if tuple(t.shape) != shape:
                raise ValueError(f"parameter shape {tuple(t.shape)} != {shape}")
        self.enti_transfer_matrix_list = [dev(params[f"enti_transfer_matrix_{n}"]) for n in range(M)]
        self.enti_transfer_bias_list = [dev(params[f"enti_transfer_bias_{n}"]) for n in range(M)]
        self.user_mlp_matrix = dev(params["user_mlp_matrix"])
        self.user_mlp_bias = dev(params["user_mlp_bias"])
        # the L+1 projection matrices live in one stack so q.W_e + b_e for every level is one launch
        self._transfer_W = dev(np.stack([params[f"transfer_matrix_{e}"] for e in range(L + 1)]))
        self._transfer_b = dev(np.stack([params[f"transfer_bias_{e}"] for e in range(L + 1)]))
        self.transfer_matrix_list = [self._transfer_W[e] for e in range(L + 1)]
        self.transfer_matrix_bias = [self._transfer_b[e] for e in range(L + 1)]
        self.transform_matrix, self.transform_bias = self.transfer_matrix_list[-1], self.transfer_matrix_bias[-1]
        self.h_emb_item_mlp_matrix = dev(params["h_emb_item_mlp_matrix"])
        self.h_emb_item_mlp_bias = dev(params["h_emb_item_mlp_bias"])
        # aggregators (model.py:290 / :359)
        self._agg = {}
        self.aggregators = []
        if not a.PS_only:
            for (i, n) in aggregator_keys(a):
                tag = f"agg_{i}_{n}_"
                name = (str(i) + "_" + str(n)) if a.wide_deep else i
                agg = self.aggregator_class(
                    self.save_model_name, self.batch_size, D, name=name,
                    User_orient_rela=(self.User_orient_rela if a.wide_deep else True),
                    weights=params[tag + "weights"], bias=params[tag + "bias"],
                    urh_weights=params[tag + "urh_weights"], urh_bias=params[tag + "urh_bias"],
                    relation_emb=self.relation_emb_matrix, device=self.device)
                self._agg[(i, n)] = agg
                self.aggregators.append(agg)
        self._profile = None
        self._native_l2_state = None
        self._native_l2_ws = {}
        # projected-tables form of the fused two-level pass inside mvin_score_l2_fwd (_prj_for_l2): None = by batch size
        self.prj = {"0": False, "1": True}.get(os.environ.get("MVIN_PRJ", ""), None)
        self._prj_tables = {}                # per stream: workspace of mvin_project_tables_elems floats, rewritten by every call
        # per-entity aggregates S0 | G of those tables (_agg_for): None = whenever the projected-tables form is taken on a shape the
        # kernels take (MVIN_L2_AGG=0 / 1 overrides: 0 answers through the library too)
        self.agg = {"0": False, "1": True}.get(os.environ.get("MVIN_L2_AGG", ""), None)
        self._agg_tables = {}                # per stream: workspace of mvin_entity_aggregates_elems floats, rewritten by every call
        # folded-tail form of the native call (_fold_for): None = whenever the aggregates form is taken (MVIN_L2_FOLD=0 / 1 overrides)
        self.fold = {"0": False, "1": True}.get(os.environ.get("MVIN_L2_FOLD", ""), None)
        self._fold_ws = {}                   # per stream: workspace of mvin_fold_tables_elems floats, rewritten by every call
        # gathered form of the grouped key addressing (mvin_key_addressing_grouped_er_fwd, _ka_er_for): on request only
        self.ka_er = os.environ.get("MVIN_KA_ER", "0") == "1"
        self._ka_er_ws = {}                  # per stream: workspace of mvin_project_relations_elems floats, rewritten by every call
        # flash form of the grouped key addressing + user MLP (mvin_key_addressing_flash_fwd, _ka_flash_for): None = automatic
        # (MVIN_KA_FLASH=0 / 1 overrides), True / False
        self.ka_flash = {"0": False, "1": True}.get(os.environ.get("MVIN_KA_FLASH", ""), None)
        self._ka_flash_ws = {}               # per stream: workspace of mvin_key_addressing_flash_tables_elems floats, rewritten by every call
        # parents of the projected-tables launch in item order (_item_order_for): None = by batch size (MVIN_ITEM_ORDER=0 / 1 overrides)
        self.item_order = {"0": False, "1": True}.get(os.environ.get("MVIN_ITEM_ORDER", ""), None)
        self._distinct_hint = None           # forward_device's distinct_users of the call at hand
        self.group_min_pairs_per_user = 4    # forward_users: batch size / n_user above which pairs are grouped by user
        self.native_l2_max_batch = 65536     # above this the pass is kernel-bound: the Python schedule costs nothing
        # up to this many pairs the whole pass is ONE kernel launch (mvin_score_small_fwd: the reference's own batch sizes,
        # 512 / 1024 per sess.run); beyond it the multi-launch schedule is the faster one (measured: 58 vs 60 us at 1 024 pairs,
        # 112 vs 66 us at 2 048: a workgroup walks its pairs' dependent-load chain alone).  The entry point
        # itself takes any batch (tests drive it to 16 384).  0 or MVIN_SMALL=0 turns it off; MVIN_SMALL_MAX overrides
        self.small_max_batch = 0 if os.environ.get("MVIN_SMALL", "1") == "0" else int(os.environ.get("MVIN_SMALL_MAX", "1024"))
        self.small_group = 0                 # pairs per workgroup of that launch (0: chosen by the library)
        self._small_state = None
        self._small_static = None
        # Device-resident feeds (forward_device / forward_users): ids are NOT validated per batch by default (it costs a
        # host sync; the kernels clamp entity / user / relation ids into their tables instead of reading out of bounds).
        # True (or MVIN_CHECK_IDS=1): every batch is checked and raises IndexError like the host-feed path and like
        # the reference's CPU tf.gather (InvalidArgument).  A user_triplet_set is always checked once, when first seen.
        self.validate_device_ids = os.environ.get("MVIN_CHECK_IDS", "0") == "1"
        self._uts_ok = None
        # grouped key addressing over STATIC per-user records (mvin_build_user_records: relation buckets, tile table, clamped
        # head / tail ids of every user's ripple sets, built once per user_triplet_set tensor): None = where the library has
        # the kernel (MVIN_KA_STATIC=0 turns it off), False = never
        self.static_user_records = None
        self._uts_records = None

    USER_RECORDS_MAX_BYTES = 8 << 30
    FLASH_TABLES_MAX_BYTES = 4 << 30

    def user_records(self, uts):
        """The static per-user records of a device-resident user_triplet_set, or None when the shape / table type has no
        kernel over them.  Cached per tensor (identity + version counter), like the one-time id check."""
        if self.static_user_records is False or os.environ.get("MVIN_KA_STATIC", "1") == "0" or self.p_hop < 1:
            return None
        if not (uts.is_cuda and uts.dtype == torch.int32 and uts.is_contiguous()):
            return None
        bf = self.entity_emb_matrix.dtype == torch.bfloat16
        if not (ops.user_records_supported(self.dim, self.p_hop, self.n_memory, self.n_relation, bf)
                or (not bf and ops.key_addressing_flash_supported(self.dim, self.p_hop, self.n_memory, self.n_relation, self.n_entity))):
            return None
        c = self._uts_records
        if c is not None and c[0]() is uts and c[1] == uts._version and c[2] == (self.n_entity, self.n_relation):
            return c[3]
        if c is not None and c[0]() is None:
            self._uts_records = None     # the tensor the records were built from is gone: free them (up to 8 GiB)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the static per-user records of this user_triplet_set are not built yet: call "
                               "model.prepare(user_triplet_set) before capturing a graph")
        words = ops.user_records_len(self.p_hop, self.n_memory, self.n_relation)
        if uts.shape[0] * words * 4 > self.USER_RECORDS_MAX_BYTES:
            return None
        rec = ops.build_user_records(uts, self.p_hop, self.n_relation, self.n_entity)
        self._uts_records = (weakref.ref(uts), uts._version, (self.n_entity, self.n_relation), rec)
        return rec

    def prepare(self, user_triplet_set=None):
        """Build, eagerly and on the current stream, everything the forward would otherwise build on first use: the
        duplicate-slot encoding of the adjacency (two launches + one host sync), the relation-logit tables of the
        aggregators and -- given a device-resident ``user_triplet_set`` -- its one-time id check and static per-user records
        (an allocation of up to USER_RECORDS_MAX_BYTES + a kernel).  Called by __init__ and set_adjacency for the encoding;
        call it yourself before capturing a forward in a hipGraph or running it on a side stream."""
        mode = os.environ.get("MVIN_L2_ENC", "auto") if self.dedup is None else ("1" if self.dedup else "0")
        if self.fused is not False and mode != "0":      # (forbidden encoding: nothing to build -- 2 launches, a sync, 2 n_entity K words)
            self.encoded_adjacency()
        for agg in self.aggregators:
            if agg.User_orient_rela:
                agg.relation_scores()
        if user_triplet_set is not None:
            self._check_uts(user_triplet_set)
            self.user_records(user_triplet_set)
        return self

    def _build_train(self):
        """model.py:378-414 (loss + Adam): the backward path is a later row of the scope
        table; nothing is built here yet."""
        self.optimizer = None
        self.trainer = None   # created on the first train() call (mvin_amd.training.Trainer)
        self._train_graphs, self._train_seen = {}, None      # captured steps per batch size (train())
        self.train_graph_max_batch = 8192

    def parameters_dict(self):
        """All parameters as numpy arrays under the names of mvin_amd/params.py."""
        L = self.n_mix_hop * self.h_hop
        p = {k: getattr(self, k).float().cpu().numpy() for k in
             ("user_emb_matrix", "entity_emb_matrix", "relation_emb_matrix", "relation_emb_KGE_matrix",
              "user_mlp_matrix", "user_mlp_bias", "h_emb_item_mlp_matrix", "h_emb_item_mlp_bias")}
        for n in range(self.n_mix_hop):
            p[f"enti_transfer_matrix_{n}"] = self.enti_transfer_matrix_list[n].cpu().numpy()
            p[f"enti_transfer_bias_{n}"] = self.enti_transfer_bias_list[n].cpu().numpy()
        for e in range(L + 1):
            p[f"transfer_matrix_{e}"] = self.transfer_matrix_list[e].cpu().numpy()
            p[f"transfer_bias_{e}"] = self.transfer_matrix_bias[e].cpu().numpy()
        for (i, n), agg in self._agg.items():
            tag = f"agg_{i}_{n}_"
            for nm in ("weights", "bias", "urh_weights", "urh_bias"):
                p[tag + nm] = getattr(agg, nm).cpu().numpy()
        return p

    # ------------------------------------------------------------------ stage-wise tables
    _STWS = ("user_emb_matrix", "entity_emb_matrix", "relation_emb_matrix", "relation_emb_KGE_matrix")

    def _emb_path(self):
        if self.path is None or getattr(self.path, "emb", None) is None:
            raise ValueError("args.path.emb is not set")
        return f"{self.path.emb}_sw_para_{self.save_model_name}_parameter.npz"

    def save_pretrain_emb_fuc(self, sess=None, saver=None):
        """model.py:66-67 / train.py:43-51: persist the four ``STWS`` embedding tables."""
        np.savez(self._emb_path(), **{k: getattr(self, k).float().cpu().numpy() for k in self._STWS})

    def restore_pretrain_emb(self):
        """train.py:53-54 counterpart."""
        with np.load(self._emb_path()) as z:
            for k in self._STWS:
                getattr(self, k).copy_(torch.from_numpy(z[k]).to(self.device))   # casts to the table dtype
        self.invalidate()

    def invalidate(self):
        """Call after changing parameters in place through raw pointers (the optimizer does):
        drops every derived table (relation logits, hoisted entity tables)."""
        for agg in self.aggregators:
            agg.invalidate()
        self._hoisted = None
        self._generation = getattr(self, "_generation", 0) + 1   # captured hipGraphs of older generations are stale

    # ------------------------------------------------------------------ graph pieces
    def get_neighbors(self, seeds, levels=None):
        """model.py:243-256 on device.  Returns (entities, relations) int32 id lists."""
        L = self.n_mix_hop * self.h_hop if levels is None else levels
        return ops.expand_ids(self.adj_entity, self.adj_relation, seeds, self.n_neighbor, L, self.n_entity)

    def _lookup(self, table, ids32):
        """tf.nn.embedding_lookup of [B] ids -> [B, D]."""
        return ops.linear([table], None, table.shape[-1], ids=[ids32])

    def _key_addressing(self, user32, item32, mem_h, mem_r, mem_t, uts=None):
        """model.py:161-240 -> user_o [B,D].  ``uts``: the ripple sets of pair b are user_triplet_set[user32[b]], indexed inside
        the kernel (mvin_key_addressing_users_fwd) instead of per-pair arrays."""
        a, D, P = self.args, self.dim, self.p_hop
        B = item32.shape[0]
        nR = self.n_relation
        n_o = P + 1 if a.PS_O_ft else P
        o_cat = torch.empty((B, n_o * D), dtype=torch.float32, device=self.device)
        w_h = self.h_emb_item_mlp_matrix.view(-1) if a.PS_O_ft else None  # first D entries multiply h ([h, user], :171)
        V = None
        if P > 0:
            # V[b,r,:] = E[item_b] . R_KGE[r]   ((R h).v == h.(v R), :214-220)
            V = torch.empty((B, nR, D), dtype=torch.float32, device=self.device)
            ops.linear([self.entity_emb_matrix], self.relation_emb_KGE_matrix, D, ids=[item32], rows=B,
                       out=V, ldo=nR * D, nz=nR, w_zstride=D * D, out_zstride=D)
        if uts is not None:
            ops.key_addressing_users(self.entity_emb_matrix, V, w_h, uts, user32, P, o_cat, n_o * D, nR)
        elif self.fused and ops.key_addressing_supported(self.n_memory, D):
            ops.key_addressing(self.entity_emb_matrix, V, w_h, mem_h, mem_r, mem_t, P, o_cat, n_o * D, nR)
        else:
            slot = 0
            if a.PS_O_ft:  # :162-197, :204-206
                ops.ripple_attn(self.entity_emb_matrix, mem_h[0], None, mem_h[0], None, w_h, 1,
                                o_cat, 0, n_o * D, nR)
                slot = 1
            for hop in range(P):  # :210-230
                ops.ripple_attn(self.entity_emb_matrix, mem_h[hop], mem_r[hop], mem_t[hop], V, None, 0,
                                o_cat, (slot + hop) * D, n_o * D, nR)
        # :232-236
        return ops.linear([o_cat], self.user_mlp_matrix, D, bias=self.user_mlp_bias)

    def _key_addressing_shared(self, item32, mem_h, mem_r, mem_t):
        """model.py:161-240 when EVERY pair of the batch carries the same ripple sets (one user
        scored against many items: the top-K evaluation of util.py:145-181).  ``mem_*``: per hop
        ONE [n_memory] int32 id list.  With the memories fixed the reads turn into dense products
            A[m] = R_KGE[r_m] . E[h_m]         ([Nm, D], once per call)
            logits = E[items] . A^T  ->  softmax over Nm  ->  o = P . E[t]
        on the MFMA linear kernel: ~2 KB of [B, Nm] traffic per pair instead of 2*Nm gathered rows,
        and no [B, nR, D] item projection.  The h-set read (:162-197) does not depend on the item."""
        a, D, P, Nm, nR = self.args, self.dim, self.p_hop, self.n_memory, self.n_relation
        E = self.entity_emb_matrix
        B = item32.shape[0]
        n_o = P + 1 if a.PS_O_ft else P
        o_cat = torch.empty((B, n_o * D), dtype=torch.float32, device=self.device)
        slot0 = 0
        if a.PS_O_ft:
            o1 = torch.empty((1, D), dtype=torch.float32, device=self.device)
            w_h = self.h_emb_item_mlp_matrix.view(-1)
            if ops.key_addressing_supported(Nm, D):
                ops.key_addressing(E, None, w_h, [mem_h[0].view(1, Nm)], [], [], 0, o1, D, nR)
            else:
                ops.ripple_attn(E, mem_h[0].view(1, Nm), None, mem_h[0].view(1, Nm), None, w_h, 1, o1, 0, D, nR)
            o_cat.view(B, n_o, D)[:, 0, :] = o1           # broadcast (plumbing)
            slot0 = 1
        if P > 0:
            Rt = self.relation_emb_KGE_matrix.transpose(1, 2).contiguous()   # W[z] = R[z]^T so that x.W = R.x
            ar = torch.arange(Nm, dtype=torch.int64, device=self.device)
        for hop in range(P):
            Hall = ops.linear([E], Rt, D, ids=[mem_h[hop]], rows=Nm, nz=nR, w_zstride=D * D)     # [nR, Nm, D]
            sel = (mem_r[hop].long() * Nm + ar).to(torch.int32)
            A = ops.linear([Hall.view(nR * Nm, D)], None, D, ids=[sel])                           # [Nm, D]
            logits = ops.linear([E], A.t().contiguous(), Nm, ids=[item32])                        # [B, Nm]
            Pm = ops.row_softmax(logits)
            T = ops.linear([E], None, D, ids=[mem_t[hop]])                                        # [Nm, D] fp32
            ops.linear([Pm], T, D, out=o_cat, out_offset=(slot0 + hop) * D, ldo=n_o * D)
        return ops.linear([o_cat], self.user_mlp_matrix, D, bias=self.user_mlp_bias)

    def _project_levels(self, ents, q, top, need_c=()):
        """model.py:267-283 for levels 0..top-1 (deeper levels are fused into the gather
        kernels).  Returns (ev list, c) where c[e] = q.W_e + b_e ([B, D]) for the levels in
        ``need_c`` and for 1..top-1 (level 0 adds q before the matmul instead)."""
        a, D, K = self.args, self.dim, self.n_neighbor
        B = ents[0].shape[0]
        E = self.entity_emb_matrix
        if not a.User_orient:
            return [ops.linear([E], None, D, ids=[ents[e].view(-1)]).view(B, -1, D) for e in range(top)], {}
        levels = sorted(set(range(1, top)) | set(need_c))
        c = {}
        if levels:
            lo, hi = levels[0], levels[-1] + 1
            cs = ops.linear([q], self._transfer_W[lo:hi], D, bias=self._transfer_b[lo:hi], nz=hi - lo,
                            w_zstride=D * D, bias_zstride=D, out_zstride=B * D)
            cs = cs.view(hi - lo, B, D)
            c = {e: cs[e - lo] for e in range(lo, hi)}
        # level 0: (E[item] + q) . W_0 + b_0 in one launch
        ev = [ops.linear([E, q], self.transfer_matrix_list[0], D, ids=[ents[0].view(-1), None],
                         bias=self.transfer_matrix_bias[0], sum_sources=True).view(B, 1, D)]
        ev += [ops.linear([E], self.transfer_matrix_list[e], D, ids=[ents[e].view(-1)], rowbias=c[e],
                          rows_per_group=K ** e).view(B, -1, D) for e in range(1, top)]
        return ev, c

    def _apply(self, agg, ev, ents, rels, hop, c, fused_level, want_probs):
        """One aggregator application at one hop (model.py:296-305)."""
        a, D, K = self.args, self.dim, self.n_neighbor
        B = ev[hop].shape[0]
        N = K ** hop
        t = agg.relation_scores() if agg.User_orient_rela else None
        wp = want_probs and agg.User_orient_rela
        if fused_level is not None:
            Wc = self.transfer_matrix_list[fused_level] if a.User_orient else None
            cc = c[fused_level] if a.User_orient else None
            if self._profile is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            res = ops.gather_attn(self.entity_emb_matrix, self.adj_entity, self.adj_relation,
                                  ents[hop].view(-1), t, ev[hop].view(B * N, D), Wc, cc, agg.weights,
                                  agg.bias, B, N, K, D, want_probs=wp)
            if self._profile is not None:
                ev1.record()
                self._profile.append((ev0, ev1))
            return res
        return ops.agg(ev[hop].view(B * N, D), ev[hop + 1].view(B * N * K, D), rels[hop].view(-1), t,
                       agg.weights, agg.bias, B, N, K, D, want_probs=wp)

    # ------------------------------------------------------------------ entity-table (hoisted) mode
    def hoist_supported(self):
        a = self.args
        return bool(a.wide_deep and not a.PS_only and self.h_hop >= 2 and self.n_neighbor <= 256)

    def _hoist_key(self):
        a0 = self._agg[(0, 0)]
        ts = (self._transfer_W, self._transfer_b, a0.weights, a0.bias, a0.urh_weights,
              self.relation_emb_matrix, self.adj_entity, self.adj_relation)
        # the entity table: identity + torch version counter, unless its owner writes it behind torch's back
        # (raw-pointer HIP kernels / collectives of dist.ShardedMVIN) and hands out a content token instead
        tok = self._table_token
        ent = (id(self.entity_emb_matrix), self.entity_emb_matrix._version) if tok is None else ("token", tok)
        return (ent,) + tuple((id(t), t._version) for t in ts)

    def hoist_entity_tables(self):
        """Entity-table mode of the two deepest levels (SURVEY.md 7.3-c route 2b; an exact
        re-association of model.py:295-305 + aggregators.py:98-146, not a different model).

        Aggregator (0,0)'s attention weights do not depend on the pair (7.3-a) and everything it
        does below level L-2 before its ReLU is linear in entity rows, so with
            S[e]  = (1/K) sum_k p_k(e) E[adj(e,k)]                 (one gather over ALL entities)
            R1[e] = (E[e].W_{L-1} + S[e].W_L) . A0                  N0[e] = S[e].W_{L-1}
        the per-pair work at a level-(L-2) node x shrinks from K + K^2 row gathers to K + 1:
            neighbors_agg for aggregator (0,0) = N0[x] + c_{L-1}/K        (c/K -> c without the softmax)
            neighbors_agg for aggregator (1,0) = (1/K) sum_k p'_k(x) relu(R1[adj(x,k)] + d)
            d = (c_{L-1} + c_L/K).A0 + a0,  c_e = q.W_e + b_e       (per pair)
        Building costs one pass over n_entity*K rows (~n_entity/K pairs' worth), so it pays as
        soon as a batch holds more level-(L-1) nodes than there are entities; it changes the
        algorithmic bytes per pair and is therefore always reported as its own mode."""
        if not self.hoist_supported():
            raise ValueError("entity-table mode needs wide_deep, h_hop >= 2 and fan-out <= 256")
        D, K = self.dim, self.n_neighbor
        L = self.n_mix_hop * self.h_hop
        E = self.entity_emb_matrix
        a0 = self._agg[(0, 0)]
        h = SimpleNamespace(key=self._hoist_key())
        t0 = a0.relation_scores() if a0.User_orient_rela else None
        S = ops.gather_mix(E, self.adj_entity, self.adj_relation, None, t0, None, self.n_entity, 1, K,
                           self.n_relation)
        if self.args.User_orient:
            W1, W2 = self.transfer_matrix_list[L - 1], self.transfer_matrix_list[L]
            b1, b2 = self.transfer_matrix_bias[L - 1], self.transfer_matrix_bias[L]
            X = ops.linear([E, S], torch.cat([W1, W2]).contiguous(), D)           # E.W1 + S.W2
            h.R1 = ops.linear([X], a0.weights, D)
            h.N0 = ops.linear([S], W1, D)
            # q -> (c1 + c2/K, c1/K) in one launch
            inv = 1.0 / K if a0.User_orient_rela else 1.0   # sum_k w_k / K: softmax weights sum to 1, plain-mean weights to K
            h.Wq = torch.stack([W2, W1]).contiguous()
            h.bq = torch.stack([b2, b1]).contiguous()
            ops.axpby(1.0, W1, inv, h.Wq[0])
            ops.axpby(1.0, b1, inv, h.bq[0])
            ops.axpby(0.0, W1, inv, h.Wq[1])
            ops.axpby(0.0, b1, inv, h.bq[1])
        else:
            h.R1 = ops.linear([E, S], a0.weights, D, sum_sources=True)
            h.N0 = S
        self._hoisted = h
        return h

    def _hoisted_naggs(self, parents, q, B):
        """(neighbors_agg of aggregator (0,0), of aggregator (1,0)) at the level-(L-2) nodes."""
        D, K = self.dim, self.n_neighbor
        h = self._hoisted
        if self.hoist == "step" or h is None or h.key != self._hoist_key():
            h = self.hoist_entity_tables()
        a0, a1 = self._agg[(0, 0)], self._agg[(1, 0)]
        P = parents.numel()
        ppp = P // B
        t1 = a1.relation_scores() if a1.User_orient_rela else None
        if self.args.User_orient:
            cs = ops.linear([q], h.Wq, D, bias=h.bq, nz=2, w_zstride=D * D, bias_zstride=D, out_zstride=B * D)
            cs = cs.view(2, B, D)
            d = ops.linear([cs[0]], a0.weights, D, bias=a0.bias)
            n0 = ops.linear([h.N0], None, D, ids=[parents], rowbias=cs[1], rows_per_group=ppp)
            n1 = ops.gather_mix(h.R1, self.adj_entity, self.adj_relation, parents, t1, d, P, ppp, K,
                                self.n_relation, relu=True)
        else:
            n0 = ops.linear([h.N0], None, D, ids=[parents])
            n1 = ops.gather_mix(h.R1, self.adj_entity, self.adj_relation, parents, t1, a0.bias.view(1, D), P, P, K,
                                self.n_relation, relu=True)
        return n0, n1

    def aggregate_delta_whole(self, item32, q, user_o, want_probs=False):
        """model.py:259-324 -> (item_embeddings [B,D], scores, sigmoid, importance_list).

        Fast path (h_hop >= 2, dim in {16,32,64,128}, fan-out a power of two): the two deepest
        levels run in ONE kernel (mvin_gather_attn_l2_fwd) that returns, per level-(L-2) node,
        the neighbor aggregates consumed by aggregator (0,0) and (1,0) at hop L-2; levels L-1
        and L are never materialised.  Otherwise the per-level kernels are used."""
        D, K, H, M = self.dim, self.n_neighbor, self.h_hop, self.n_mix_hop
        L = M * H
        B = item32.shape[0]
        use_hoist = bool(self.hoist) and not want_probs and self.hoist_supported()
        use_l2 = use_hoist or (self.fused and H >= 2 and ops.gather_attn_l2_supported(D, K))
        top = L - 1 if use_l2 else L          # levels 0..top-1 are materialised
        # depth-2 trees: everything above the fused kernel (level-0 projection, both hop-0 aggregators, the
        # combiner and the score) is ONE launch (mvin_l2_tail_fwd) instead of four
        use_tail = (use_l2 and not use_hoist and L == 2 and M == 1 and self.fused is not False
                    and ops.l2_tail_supported(D))
        if use_tail and item32.dtype == torch.int64:
            # the parents of a depth-2 tree are the items themselves: the fused kernel reads the int64 ids in place
            # (mvin_gather_attn_l2_fwd_i64), no level-0 id list is written
            ents, rels = [item32], []
        else:
            ents, rels = self.get_neighbors(item32, levels=top - 1)
        ev, c = (None, {}) if use_tail else self._project_levels(ents, q, top, need_c=() if use_l2 else (L,))
        nagg = pp = pc = None
        if use_hoist:
            if self._profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            nagg = self._hoisted_naggs(ents[L - 2].view(-1), q, B)
            if self._profile is not None:
                e1.record()
                self._profile.append((e0, e1))
        elif use_l2:
            a0, a1 = self._agg[(0, 0)], self._agg[(1, 0)]
            uo = self.args.User_orient
            enc = self._enc_for_l2(want_probs, n_parents=B * K ** (L - 2))
            tabs = None
            prj_now = not want_probs and (enc is not None or self._prj_plain_ok()) and self._prj_for_l2(B, B * K ** (L - 2))
            if prj_now and use_tail and self._fold_for(enc) and self.entity_emb_matrix.dtype == torch.float32:
                # folded-tail form (mvin_fold_tables -> mvin_score_l2_folded_fwd): tables and aggregates from the current parameters,
                # then everything above key addressing in one launch -- what mvin_score_l2_fwd runs for this call
                Wp, bp = self.transfer_matrix_list, self.transfer_matrix_bias
                cs = torch.cuda.current_stream().cuda_stream
                n_ws = ops._lib.load().mvin_fold_tables_elems(self.n_entity, D)
                fw = self._fold_ws.get(cs)
                if fw is None or fw.numel() != n_ws:
                    fw = self._fold_ws[cs] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
                t0 = a0.relation_scores() if a0.User_orient_rela else None
                t1 = a1.relation_scores() if a1.User_orient_rela else None
                if self._profile is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ops.fold_tables(self.entity_emb_matrix, enc[0], enc[1], t0, Wp[0], bp[0], Wp[1], bp[1], Wp[2], bp[2], a0.weights, a0.bias,
                                self.enti_transfer_matrix_list[0], self.enti_transfer_bias_list[0], a1.weights, K, self.n_relation, out=fw)
                item_emb, scores, sig = ops.score_l2_folded(fw, enc[0], enc[1], item32, t0, t1, q, user_o, a1.weights, a1.bias,
                                                            self.enti_transfer_matrix_list[0], K, D, self.n_relation, self.n_entity)
                if self._profile is not None:
                    e1.record()
                    self._profile.append((e0, e1))
                return item_emb, scores, sig, []
            # (the projected-tables kernels write no attention outputs: a want_probs pass keeps the form that does)
            if prj_now:
                # projected-tables form (mvin_gather_attn_l2_prj_fwd): E.W1 | E.W1.A0 | E.W2.A0 from the current parameters, per call
                Wp, bp = self.transfer_matrix_list, self.transfer_matrix_bias
                tabs = ops.project_tables(self.entity_emb_matrix, Wp[L - 1], Wp[L], bp[L - 1], bp[L], a0.weights, a0.bias, K,
                                          a0.User_orient_rela)
            if self._profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            l2_args = (a0.relation_scores() if a0.User_orient_rela else None,
                       a1.relation_scores() if a1.User_orient_rela else None,
                       self.transfer_matrix_list[L - 1] if uo else None, self.transfer_matrix_list[L] if uo else None,
                       self.transfer_matrix_bias[L - 1] if uo else None, self.transfer_matrix_bias[L] if uo else None,
                       q if uo else None, a0.weights, a0.bias, B, K ** (L - 2), K, D, self.n_relation)
            if tabs is not None:
                ae, ar = (enc[0], enc[1]) if enc is not None else (self.adj_entity, self.adj_relation)
                order = None
                if self._agg_for(enc):
                    # per-entity aggregates S0 | G of the tables just built: ~cnt rows per parent instead of ~cnt^2
                    cs = torch.cuda.current_stream().cuda_stream
                    n_ws = ops._lib.load().mvin_entity_aggregates_elems(self.n_entity, D)
                    at = self._agg_tables.get(cs)
                    if at is None or at.numel() != n_ws:
                        at = self._agg_tables[cs] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
                    ops.entity_aggregates(tabs, ae, ar, l2_args[0], K, D, self.n_relation, self.n_entity, out=at)
                    n0, n1 = ops.gather_attn_l2_agg(tabs, at, ae, ar, ents[L - 2].view(-1), l2_args[0], l2_args[1], q, B,
                                                    K ** (L - 2), K, D, self.n_relation, self.n_entity)
                else:
                    if enc is not None and L == 2 and self._item_order_for(B) and ops.gather_attn_l2_wpp_supported(D, K):
                        order = ops.order_by_key(ents[0].view(-1))
                    n0, n1 = ops.gather_attn_l2_prj(tabs, ae, ar, ents[L - 2].view(-1), l2_args[0], l2_args[1], q, B,
                                                    K ** (L - 2), K, D, self.n_relation, self.n_entity, encoded=enc is not None, order=order)
                pp = pc = None
            elif enc is not None:
                n0, n1 = ops.gather_attn_l2_enc(self.entity_emb_matrix, enc[0], enc[1], ents[L - 2].view(-1), *l2_args)
                pp = pc = None
            else:
                n0, n1, pp, pc = ops.gather_attn_l2(
                    self.entity_emb_matrix, self.adj_entity, self.adj_relation, ents[L - 2].view(-1), *l2_args,
                    want_probs=want_probs and a0.User_orient_rela)
            if self._profile is not None:
                e1.record()
                self._profile.append((e0, e1))
            nagg = (n0, n1)
            if use_tail:
                item_emb, scores, sig = ops.l2_tail(
                    self.entity_emb_matrix, item32, q if uo else None, user_o, n0, n1,
                    self.transfer_matrix_list[0] if uo else None, self.transfer_matrix_bias[0] if uo else None,
                    a0.weights, a0.bias, a1.weights, a1.bias, self.enti_transfer_matrix_list[0],
                    self.enti_transfer_bias_list[0])
                importance = []
                if want_probs and a0.User_orient_rela:
                    importance = [pp.view(B, 1, K) if pp is not None else None,
                                  pc.view(B, K, K) if pc is not None else None]
                elif want_probs:
                    importance = [None, None]
                return item_emb, scores, sig, importance
        importance = []
        out = None
        for n in range(M):
            stages = [ev]
            for i in range(H):
                agg = self._agg[(i, n)]
                nxt, probs = [], []
                for hop in range(L - (H * n + i)):
                    if use_l2 and n == 0 and i == 0 and hop == L - 1:
                        nxt.append(None)  # consumed inside the fused kernel (as nagg[1])
                        probs.append(pc.view(B, K ** (L - 1), K) if pc is not None else None)
                        continue
                    if use_l2 and n == 0 and i <= 1 and hop == L - 2:
                        # aggregators.py:108-116 with neighbors_agg from the fused kernel
                        o = ops.linear([ev[hop].view(-1, D), nagg[i]], agg.weights, D, bias=agg.bias,
                                       relu=True, sum_sources=True).view(B, -1, D)
                        nxt.append(o)
                        probs.append(pp.view(B, K ** (L - 2), K) if (i == 0 and pp is not None) else None)
                        continue
                    fused = L if (not use_l2 and n == 0 and i == 0 and hop == L - 1) else None
                    o, p = self._apply(agg, ev, ents, rels, hop, c, fused, want_probs and i == 0)
                    nxt.append(o)
                    probs.append(p)
                if i == 0:
                    importance = probs
                ev = nxt
                stages.append(ev)
            keep = (M - n - 1) * H + 1
            last = n == M - 1
            new = []
            for e in range(keep):  # :310-315
                res = ops.linear([st[e].view(-1, D) for st in stages], self.enti_transfer_matrix_list[n], D,
                                 bias=self.enti_transfer_bias_list[n], score_u=user_o if last else None)
                if last:
                    out = res
                else:
                    new.append(res.view(B, -1, D))
            ev = new
        item_emb, scores, sig = out
        return item_emb, scores, sig, importance

    def aggregate(self, item32, q, user_o, want_probs=False):
        """model.py:327-376 (wide_deep=False).  The reference revision cannot run this path
        (it treats the aggregator's tuple as a tensor, :366-374); the evident intent --
        element [0] -- is implemented.  Untested by the reference."""
        D, H = self.dim, self.h_hop
        ents, rels = self.get_neighbors(item32, levels=H - 1)
        ev, c = self._project_levels(ents, q, H, need_c=(H,))
        for i in range(H):
            agg = self._agg[(i, 0)]
            nxt = []
            for hop in range(H - i):
                fused = H if (i == 0 and hop == H - 1) else None
                o, _ = self._apply(agg, ev, ents, rels, hop, c, fused, False)
                nxt.append(o)
            ev = nxt
        item_emb, scores, sig = ops.linear([ev[0].view(-1, D)], None, D, score_u=user_o)
        return item_emb, scores, sig, []

    # ------------------------------------------------------------------ forward
    def forward_users(self, user_indices, item_indices, user_triplet_set, want_probs=False, distinct_users=None):
        """The same pass when the caller holds ``user_triplet_set`` on the device ([n_user, max(1,P), 3, n_memory]
        int32, data_loader_user_set.py:392-441) instead of per-pair ripple-set arrays: the feed assembly of
        train.py:117-120 (memories_x[i] = user_triplet_set[user][i][x]) happens inside the key-addressing kernel,
        which groups the batch's pairs by user and reads each user's rows once (``_key_addressing_grouped``).
        ``distinct_users``: how many different users the caller expects in THIS batch when it knows better than
        "any of n_user" (a rank that scores the pairs of 1/W of the users; a top-K evaluation over 250 users): the
        static pairs-per-user rule that picks the grouped form uses it instead of n_user."""
        return self.forward_device(user_indices, item_indices, None, None, None, want_probs=want_probs,
                                   uts=user_triplet_set, distinct_users=distinct_users)

    # ------------------------------------------------------------------ native whole-pass schedule
    def _native_l2_ok(self, item, memories_h, want_probs, cap=True):
        """The pass can be enqueued by ONE native call (mvin_score_l2_fwd): default wiring, depth-2 trees.
        ``memories_h`` None = the users feed (user_triplet_set + user ids)."""
        a = self.args
        return (self.fused is not False and not want_probs and not self.hoist and self._profile is None
                and a.wide_deep and not a.PS_only and not a.HO_only and a.User_orient_kg_eh
                and self.n_mix_hop == 1 and self.h_hop == 2 and self.p_hop >= 1
                and item.dtype == torch.int64 and (memories_h is None or memories_h[0].dim() == 2)
                and ops.l2_tail_supported(self.dim) and ops.gather_attn_l2_supported(self.dim, self.n_neighbor)
                and ops.key_addressing_supported(self.n_memory, self.dim)
                and (not cap or item.shape[0] <= self.native_l2_max_batch))

    def _small_static_ok(self):
        """Everything about ``_small_ok`` that does not depend on the batch, cached per parameter generation (the check
        runs on every call of a path whose whole GPU time is ~20 us)."""
        key = (self._generation, self.fused, bool(self.hoist), self.entity_emb_matrix.data_ptr())
        c = self._small_static
        if c is None or c[0] != key:
            a = self.args
            ok = (self.fused is not False and not self.hoist and a.wide_deep and not a.PS_only and not a.HO_only
                  and a.User_orient_kg_eh and self.n_mix_hop == 1 and self.h_hop in (1, 2) and 1 <= self.p_hop <= 4
                  and self.entity_emb_matrix.dtype == torch.float32
                  and self.entity_emb_matrix.numel() * 4 < (1 << 32) - 4096 and self.adj_entity.numel() * 4 < (1 << 32) - 4096
                  and ops.score_small_supported(self.dim, self.n_neighbor, self.p_hop, self.n_memory, self.n_relation))
            c = self._small_static = (key, bool(ok))
        return c[1]

    def _small_ok(self, item, memories_h, want_probs):
        """The pass can be ONE kernel launch (mvin_score_small_fwd): default wiring, depth-2 trees, fp32 table below 4 GiB,
        at most ``small_max_batch`` pairs.  ``memories_h`` None = the users feed."""
        return (item.shape[0] <= self.small_max_batch and not want_probs and self._profile is None
                and item.dtype == torch.int64 and (memories_h is None or memories_h[0].dim() == 2) and self._small_static_ok())

    def _score_small_native(self, item, mem_h, mem_r, mem_t, uts=None, users=None):
        """model.py:125-159 as ONE kernel launch (mvin_score_small_fwd).  The argument block is built once per parameter
        generation; per call only the table / logit-table / feed / output pointers are refreshed and ONE output buffer is
        allocated (the four result tensors are views made when somebody reads them: the host side of this path is as long
        as its kernel)."""
        from . import _lib
        import ctypes as C
        D, P, B = self.dim, self.p_hop, item.shape[0]
        a0, a1 = self._agg[(0, 0)], self._agg.get((1, 0))          # a one-hop tree (h_hop = 1) has aggregator (0,0) only
        t0 = a0.relation_scores() if a0.User_orient_rela else None
        t1 = a1.relation_scores() if (a1 is not None and a1.User_orient_rela) else None
        st = self._small_state
        if st is None or st["key"] != self._generation or st["dedup"] != self.dedup:
            a = self.args
            ptr = lambda t: t.data_ptr() if t is not None else None
            uo = a.User_orient
            s = _lib.ScoreL2Args()
            s.adj_entity, s.adj_relation = ptr(self.adj_entity), ptr(self.adj_relation)
            s.relation_kge = ptr(self.relation_emb_KGE_matrix)
            s.h_set_w = ptr(self.h_emb_item_mlp_matrix) if a.PS_O_ft else None
            s.user_mlp_W, s.user_mlp_b = ptr(self.user_mlp_matrix), ptr(self.user_mlp_bias)
            L = self.h_hop
            for e in range(3):
                setattr(s, f"W{e}", ptr(self.transfer_matrix_list[e]) if (uo and e <= L) else None)
                setattr(s, f"b{e}", ptr(self.transfer_matrix_bias[e]) if (uo and e <= L) else None)
            s.A0, s.a0 = ptr(a0.weights), ptr(a0.bias)
            s.A1, s.a1 = (ptr(a1.weights), ptr(a1.bias)) if a1 is not None else (None, None)
            s.Wmix, s.bmix = ptr(self.enti_transfer_matrix_list[0]), ptr(self.enti_transfer_bias_list[0])
            s.D, s.K, s.P, s.Nm, s.depth = D, self.n_neighbor, P, self.n_memory, L
            s.n_entity, s.n_relation, s.table_bf16 = self.n_entity, self.n_relation, 0
            # the duplicate-slot encoding whenever the adjacency has one (distinct rows only; a row without repeats costs the
            # same either way) unless the plain adjacency is forced (MVIN.dedup = False / MVIN_L2_ENC=0)
            mode = os.environ.get("MVIN_L2_ENC", "auto") if self.dedup is None else ("1" if self.dedup else "0")
            enc = self.encoded_adjacency() if mode != "0" else None
            s.enc_entity, s.enc_relation = (ptr(enc[0]), ptr(enc[1])) if enc is not None else (None, None)
            keep = (self.adj_entity, self.adj_relation, self.relation_emb_KGE_matrix, self.h_emb_item_mlp_matrix,
                    self.user_mlp_matrix, self.user_mlp_bias, list(self.transfer_matrix_list), list(self.transfer_matrix_bias),
                    a0.weights, a0.bias, a1.weights if a1 is not None else None, a1.bias if a1 is not None else None,
                    self.enti_transfer_matrix_list[0], self.enti_transfer_bias_list[0], enc)
            st = self._small_state = {"key": self._generation, "args": s, "ref": C.byref(s), "keep": keep, "arr": (C.c_void_p * P),
                                      "dedup": self.dedup, "fn": _lib.load().mvin_score_small_fwd, "live": None}
        s = st["args"]
        E = self.entity_emb_matrix
        s.entity_emb = E.data_ptr()
        s.t0 = t0.data_ptr() if t0 is not None else None
        s.t1 = t1.data_ptr() if t1 is not None else None
        out = torch.empty((2 * B * D + 2 * B,), dtype=torch.float32, device=self.device)      # one allocation for the four outputs
        s.items = item.data_ptr()
        if uts is not None:
            s.uts, s.users, s.n_user = uts.data_ptr(), users.data_ptr(), uts.shape[0]
            s.mem_h = s.mem_r = s.mem_t = None
        else:
            arr = st["arr"]
            hold = (arr(*[t.data_ptr() for t in mem_h[:P]]), arr(*[t.data_ptr() for t in mem_r[:P]]),
                    arr(*[t.data_ptr() for t in mem_t[:P]]))
            st["live"] = hold                 # (read by the call below only; kept until the next call replaces it)
            s.uts = s.users = None
            s.mem_h, s.mem_r, s.mem_t = C.addressof(hold[0]), C.addressof(hold[1]), C.addressof(hold[2])
        base = out.data_ptr()
        s.user_o, s.item_emb, s.scores, s.sig = base, base + 4 * B * D, base + 8 * B * D, base + 8 * B * D + 4 * B
        s.B = B
        rc = st["fn"](st["ref"], self.small_group, torch.cuda.current_stream().cuda_stream)
        if rc:
            _lib.check(rc, "mvin_score_small_fwd")
        return _SmallOut(out, B, D)

    def _ka_er_for(self, uts, records):
        """Gathered form of the grouped key addressing for this call?  OFF unless asked for (``self.ka_er = True`` / MVIN_KA_ER=1):
        measured at C3 the kernel over the records goes 0.94 -> 0.74 ms, the table R_KGE[r] . E[e] (245 MB, rebuilt per call) costs
        0.11 ms, and with two scoring streams its writes compete with the other stream's kernels -- 2.74 -> 2.67 ms per step on
        one stream, 197.0 -> 197.6 M pairs/s on two."""
        a = self.args
        if not self.ka_er or records is None or self.entity_emb_matrix.dtype != torch.float32 or self.p_hop < 1:
            return False
        return bool(ops.key_addressing_grouped_er_supported(self.dim, self.p_hop, self.n_memory, self.n_relation, self.n_entity,
                                                            bool(a.PS_O_ft)))

    def _ka_flash_for(self, uts, records, B):
        """Flash form of the grouped key addressing + user MLP (mvin_key_addressing_flash_fwd) for a batch of B pairs?  Its per-call
        tables cost ~(nR + P + 1) n_entity rows of work whatever the batch (0.15 ms at BASELINE C3, where the kernel then takes 0.52
        ms against 0.89 + 0.15 for the kernel over the records + the MLP launch; amazon-book's 39 relations make it a 1.1 GB table):
        automatic when the users the batch can hold reference at least as many ripple rows as the table has -- users x P x Nm >=
        nR x n_entity.  ``self.ka_flash`` True / False (MVIN_KA_FLASH=1 / 0) forces it."""
        if (records is None or self.ka_flash is False or self.entity_emb_matrix.dtype != torch.float32 or self.p_hop < 1
                or not ops.key_addressing_flash_supported(self.dim, self.p_hop, self.n_memory, self.n_relation, self.n_entity)):
            return False
        if self.ka_flash:
            return True
        users = min(int(B), int(uts.shape[0]), int(self._distinct_hint or uts.shape[0]))
        # (and never a workspace beyond FLASH_TABLES_MAX_BYTES per stream on its own initiative: nR = 100 relations over 10^6
        #  entities would be a 27 GB table)
        if _lib_flash_elems(self) * 4 > self.FLASH_TABLES_MAX_BYTES:
            return False
        return users * self.p_hop * self.n_memory >= self.n_relation * self.n_entity

    def _ka_flash_tables(self, stream):
        n_ws = _lib_flash_elems(self)
        ws = self._ka_flash_ws.get(stream)
        if ws is None or ws.numel() != n_ws:
            ws = self._ka_flash_ws[stream] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
        return ws

    ITEM_ORDER_MIN_BATCH = 32768

    def _item_order_for(self, B):
        """Parents of the projected-tables launch in ITEM order (mvin_order_by_key, ~36 us per 524 288 pairs)?  Pairs of the same item
        gather the same rows; back to back they are cache hits -- the launch's requests past the L2 halve at BASELINE C3 (7.5 -> 3.6 GB),
        and those bytes are what bounds the step.  For the wave-per-parent kernel (dim 64, fan-out <= 32, depth-2 trees: the parents
        are the pairs) and batches large enough to hold repeated items; ``self.item_order`` True / False (MVIN_ITEM_ORDER=1 / 0) forces it."""
        if self.dim != 64 or self.n_neighbor > 32 or self.n_mix_hop * self.h_hop != 2 or os.environ.get("MVIN_L2_WPP", "1") == "0":
            return False
        want = self.item_order
        if want is None:
            want = B >= self.ITEM_ORDER_MIN_BATCH
        return bool(want)

    def _agg_for(self, enc):
        """Per-entity aggregates form of the two deepest levels (mvin_project_tables -> mvin_entity_aggregates ->
        mvin_gather_attn_l2_agg_fwd) for a call that takes the projected-tables form?  The aggregates cost ~17 gathered rows per
        entity and save a parent ~100 of its ~120 (C3): whenever the tables themselves pay (_prj_for_l2's rule is the stricter one),
        on the shapes the kernels take (D = 64, K in {16, 32, 64}, encoded adjacency).  ``self.agg`` False (MVIN_L2_AGG=0) keeps the kernels
        over the tables."""
        if enc is None or self.agg is False:
            return False
        return self._agg_shape_ok()

    def _agg_shape_ok(self):
        """mvin_gather_attn_l2_agg_supported for this model's tables (D = 64, K in {16, 32, 64}, sizes within 32-bit offsets)."""
        key = (self.n_entity, self.n_relation, self.dim, self.n_neighbor)
        c = getattr(self, "_agg_ok_cache", None)
        if c is None or c[0] != key:
            c = self._agg_ok_cache = (key, ops.gather_attn_l2_agg_supported(self.dim, self.n_neighbor, self.n_entity, self.n_relation))
        return c[1]

    def _fold_for(self, enc):
        """Folded-tail form of the native call (mvin_fold_tables -> mvin_score_l2_folded_fwd: the aggregates form with nagg0 and ev0
        folded into per-entity tables too -- six products per pair instead of eight) wherever the aggregates form is taken by
        mvin_score_l2_fwd (depth-2 trees, User_orient on).  ``self.fold`` False (MVIN_L2_FOLD=0) keeps aggregates + mvin_l2_tail_fwd."""
        if enc is None or self.fold is False or self.agg is False or not self.args.User_orient:
            return False
        return self._fold_shape_ok()

    def _fold_gather_ok(self):
        """mvin_score_l2_folded_gather_supported for this model's tables (dim 64, K in {16, 32}; MVIN_L2_FOLD_GATHER=0 says no)."""
        key = (self.n_entity, self.n_relation, self.dim, self.n_neighbor)
        c = getattr(self, "_fold_gather_cache", None)
        if c is None or c[0] != key:
            c = self._fold_gather_cache = (key, ops.score_l2_folded_gather_supported(self.dim, self.n_neighbor, self.n_entity, self.n_relation))
        return c[1]

    def _fold_shape_ok(self):
        """mvin_score_l2_folded_supported for this model's tables (dim 64 with K in {16, 32, 64} like the aggregates form, dim 32 with K in {16, 32})."""
        key = (self.n_entity, self.n_relation, self.dim, self.n_neighbor)
        c = getattr(self, "_fold_ok_cache", None)
        if c is None or c[0] != key:
            c = self._fold_ok_cache = (key, ops.score_l2_folded_supported(self.dim, self.n_neighbor, self.n_entity, self.n_relation))
        return c[1]

    def _prj_plain_ok(self):
        """The projected-tables form over the PLAIN adjacency: the wave-per-parent kernel of D = 32, K in {8, 16} (BASELINE C2) --
        where the library takes THAT kernel for this model's tables (mvin_gather_attn_l2_prj_supported: its LDS copy of the
        relation logits fits, the adjacency is below 2 GiB, MVIN_L2_D32 does not forbid it); otherwise the unprojected form stays."""
        if not (self.dim == 32 and self.n_neighbor in (8, 16) and self.fused is not False):
            return False
        key = (self.n_entity, self.n_relation, self.entity_emb_matrix.dtype)
        c = getattr(self, "_prj_plain_cache", None)
        if c is None or c[0] != key:
            ok = (self.entity_emb_matrix.dtype == torch.float32
                  and ops.gather_attn_l2_prj_supported(self.dim, self.n_neighbor, False, self.n_entity, self.n_relation))
            c = self._prj_plain_cache = (key, ok)
        return c[1]

    def _prj_for_l2(self, B, n_parents=None):
        """Projected-tables form of the fused two-level pass for a batch of B pairs (``n_parents`` level-(L-2) nodes)?
        ``self.prj``: None = automatic (MVIN_PRJ=0 / 1 overrides), True / False."""
        if not (self.args.User_orient and self.entity_emb_matrix.dtype == torch.float32
                and self.entity_emb_matrix.numel() * 4 < (1 << 30) and B * self.dim * 4 < (1 << 30)):
            return False
        want = self.prj
        if want is None:
            # D <= 64, K <= 32: the other instances of the kernel are at their register budget already (K = 64: the second self row
            # costs 18 spilled registers in the front role's id pipeline) and measured no faster (C4) -- on request only
            # (K = 64 takes it where the per-entity aggregates exist, D = 64: there the tables are only the aggregates' input)
            # With the aggregates / the folded tail behind the tables the form pays earlier -- measured break-even (pairs per step, one GPU):
            # C3 (K = 32) ~32 768 = 10 n_entity / K, C4 (K = 64) ~8 192 = 4.6 n_entity / K; the kernels over the tables themselves: 16
            aggs = self.agg is not False and ((self.dim == 64 and self._agg_shape_ok())
                                              or (self.dim == 32 and self.n_mix_hop * self.h_hop == 2 and self.fold is not False
                                                  and self._fold_shape_ok()))
            k_ok = self.n_neighbor <= 32 or (self.n_neighbor == 64 and aggs)
            factor = (5 if self.n_neighbor == 64 else 10) if aggs else 16
            want = self.dim <= 64 and k_ok and (n_parents or B) * self.n_neighbor >= factor * self.n_entity
        return bool(want)

    def _score_l2_native(self, item, mem_h, mem_r, mem_t, uts=None, users=None, grouped=False):
        """model.py:125-159 through mvin_score_l2_fwd.  The argument block (every weight pointer) is built once and
        kept until a parameter tensor is replaced; per call only the batch pointers change."""
        from . import _lib
        import ctypes as C
        a, D, P, B = self.args, self.dim, self.p_hop, item.shape[0]
        a0, a1 = self._agg[(0, 0)], self._agg[(1, 0)]
        t0 = a0.relation_scores() if a0.User_orient_rela else None
        t1 = a1.relation_scores() if a1.User_orient_rela else None
        uo = a.User_orient
        # The block is rebuilt when a parameter tensor may have been REPLACED (generation); what legitimately changes
        # from call to call -- the entity table (ShardedMVIN alternates two working tables), the relation-logit tables
        # (rebuilt after every optimizer step) and the batch -- is refreshed per call, and the workspaces live outside
        # the block, per (batch size, stream): no allocation churn on the small-batch path this call exists for, and
        # two streams scoring the same batch size never share V / o_cat / nagg.
        key = self._generation
        ptr = lambda t: t.data_ptr() if t is not None else None
        st = self._native_l2_state
        if st is None or st["key"] != key:
            s = _lib.ScoreL2Args()
            s.adj_entity, s.adj_relation = ptr(self.adj_entity), ptr(self.adj_relation)
            s.relation_kge = ptr(self.relation_emb_KGE_matrix)
            s.h_set_w = ptr(self.h_emb_item_mlp_matrix) if a.PS_O_ft else None
            s.user_mlp_W, s.user_mlp_b = ptr(self.user_mlp_matrix), ptr(self.user_mlp_bias)
            for e in range(3):
                setattr(s, f"W{e}", ptr(self.transfer_matrix_list[e]) if uo else None)
                setattr(s, f"b{e}", ptr(self.transfer_matrix_bias[e]) if uo else None)
            s.A0, s.a0, s.A1, s.a1 = ptr(a0.weights), ptr(a0.bias), ptr(a1.weights), ptr(a1.bias)
            s.Wmix, s.bmix = ptr(self.enti_transfer_matrix_list[0]), ptr(self.enti_transfer_bias_list[0])
            s.D, s.K, s.P, s.Nm = D, self.n_neighbor, P, self.n_memory
            s.n_entity, s.n_relation = self.n_entity, self.n_relation
            # every tensor whose address sits in the block is kept alive with it; rebinding a parameter attribute
            # needs invalidate(), exactly as for the cached relation logits
            keep = (self.adj_entity, self.adj_relation, self.relation_emb_KGE_matrix,
                    self.h_emb_item_mlp_matrix, self.user_mlp_matrix, self.user_mlp_bias, list(self.transfer_matrix_list),
                    list(self.transfer_matrix_bias), a0.weights, a0.bias, a1.weights, a1.bias,
                    self.enti_transfer_matrix_list[0], self.enti_transfer_bias_list[0])
            st = self._native_l2_state = {"key": key, "args": s, "keep": keep, "arr": (C.c_void_p * P)}
        s = st["args"]
        s.entity_emb, s.t0, s.t1 = ptr(self.entity_emb_matrix), ptr(t0), ptr(t1)
        s.table_bf16 = 1 if self.entity_emb_matrix.dtype == torch.bfloat16 else 0
        enc = self._enc_for_l2(n_parents=B)
        s.enc_entity, s.enc_relation = (ptr(enc[0]), ptr(enc[1])) if enc is not None else (None, None)
        rec = self.user_records(uts) if grouped else None
        s.user_records = ptr(rec)
        st["live"] = (self.entity_emb_matrix, t0, t1, enc, rec)
        s.ka_er = s.ka_flash = None
        if grouped and self._ka_flash_for(uts, rec, B):    # flash form: its tables are rebuilt by every call
            s.ka_flash = self._ka_flash_tables(torch.cuda.current_stream().cuda_stream).data_ptr()
        elif grouped and self._ka_er_for(uts, rec):        # gathered U rows: the table is rebuilt by every call
            cs = torch.cuda.current_stream().cuda_stream
            n_ws = _lib.load().mvin_project_relations_elems(self.n_entity, self.n_relation, D)
            ew = self._ka_er_ws.get(cs)
            if ew is None or ew.numel() != n_ws:
                ew = self._ka_er_ws[cs] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
            s.ka_er = ew.data_ptr()
        n_o = P + (1 if a.PS_O_ft else 0)
        stream = torch.cuda.current_stream()
        # projected-tables form of the two deepest levels (mvin_gather_attn_l2_prj_fwd): E.W1 | E.W1.A0 | E.W2.A0 is rebuilt by every call
        # from the current parameters -- worth it when the batch's distinct children outnumber the entities (the per-entity
        # products cost ~n_entity rows of work, the per-child products they replace ~B K / 4)
        prj = (enc is not None or self._prj_plain_ok()) and self._prj_for_l2(B)
        fold = bool(prj and self._fold_for(enc))           # (its own workspace holds its own tables)
        # MVIN.agg = False ("every pair gathers its own rows"): the folded tail still applies, in its gather form -- per-row tables only
        gfold = bool(prj and not fold and enc is not None and self.agg is False and self.fold is not False and self.args.User_orient
                     and self.n_mix_hop * self.h_hop == 2 and self._fold_gather_ok())
        if prj and not fold and not gfold:
            pt = self._prj_tables.get(stream.cuda_stream)
            n_ws = _lib.load().mvin_project_tables_elems(self.n_entity, D)
            if pt is None or pt.numel() != n_ws:
                pt = self._prj_tables[stream.cuda_stream] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
            s.prj_tables = pt.data_ptr()
        else:
            s.prj_tables = None
        s.agg_tables = s.fold_ws = None
        s.fold_gather = 0
        if fold or gfold:
            fw = self._fold_ws.get(stream.cuda_stream)
            n_ws = _lib.load().mvin_fold_tables_elems(self.n_entity, D)
            if fw is None or fw.numel() != n_ws:
                fw = self._fold_ws[stream.cuda_stream] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
            s.fold_ws = fw.data_ptr()
            s.fold_gather = 1 if gfold else 0
        elif prj and self._agg_for(enc):
            at = self._agg_tables.get(stream.cuda_stream)
            n_ws = _lib.load().mvin_entity_aggregates_elems(self.n_entity, D)
            if at is None or at.numel() != n_ws:
                at = self._agg_tables[stream.cuda_stream] = torch.empty((n_ws,), dtype=torch.float32, device=self.device)
            s.agg_tables = at.data_ptr()
        wkey = (B, n_o, stream.cuda_stream, bool(grouped), uts.shape[0] if grouped else 0)
        ws = self._native_l2_ws.get(wkey)
        if ws is None:        # reused across calls of the same batch size ON THE SAME STREAM (which orders the reuse)
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)
            # grouped key addressing needs no [B, nR, D] item projection; its sort workspace instead
            gws = torch.empty(2 * uts.shape[0] + 4 * B + 3, dtype=torch.int32, device=self.device) if grouped else None
            ows = None                    # (the item-order workspace: allocated when first wanted, below)
            ws = self._native_l2_ws[wkey] = (None if grouped else f(B, self.n_relation, D), f(B, n_o * D),
                                             torch.empty(B, dtype=torch.int32, device=self.device), f(B, D), f(B, D), gws, ows)
            if len(self._native_l2_ws) > 8:
                self._native_l2_ws.pop(next(iter(self._native_l2_ws)))
        user_o = torch.empty((B, D), dtype=torch.float32, device=self.device)
        item_emb = torch.empty((B, D), dtype=torch.float32, device=self.device)
        scores = torch.empty((B,), dtype=torch.float32, device=self.device)
        sig = torch.empty((B,), dtype=torch.float32, device=self.device)
        s.items = item.data_ptr()
        if uts is not None:       # users feed: the kernel reads the lists of users[b] out of user_triplet_set
            s.uts, s.users, s.n_user = uts.data_ptr(), users.data_ptr(), uts.shape[0]
            s.mem_h = s.mem_r = s.mem_t = None
        else:
            ph, pr, pt = (st["arr"](*[t.data_ptr() for t in lst[:P]]) for lst in (mem_h, mem_r, mem_t))
            s.uts = s.users = None
            s.mem_h, s.mem_r, s.mem_t = C.addressof(ph), C.addressof(pr), C.addressof(pt)
        s.V, s.o_cat, s.parents, s.nagg0, s.nagg1, s.group_ws = (w.data_ptr() if w is not None else None for w in ws[:6])
        s.item_order_ws = None
        # (the aggregates form gathers ~12 rows of a 27 MB table per pair: item order buys it nothing -- measured 235 vs 252 us at C3)
        if prj and enc is not None and s.agg_tables is None and (s.fold_ws is None or gfold) and self._item_order_for(B):
            if ws[6] is None:
                ws = self._native_l2_ws[wkey] = ws[:6] + (torch.empty(_lib.load().mvin_order_by_key_ws_elems(B) + B, dtype=torch.int32,
                                                                      device=self.device),)
            s.item_order_ws = ws[6].data_ptr()
        s.user_o, s.item_emb, s.scores, s.sig = user_o.data_ptr(), item_emb.data_ptr(), scores.data_ptr(), sig.data_ptr()
        s.B = B
        _lib.check(_lib.load().mvin_score_l2_fwd(C.byref(s), C.c_void_p(stream.cuda_stream)), "mvin_score_l2_fwd")
        return SimpleNamespace(scores=scores, scores_normalized=sig, user_o=user_o, item_embeddings=item_emb,
                               importance_list=[])

    def _key_addressing_grouped(self, user, item, uts):
        """model.py:161-240 with the pairs grouped by user (mvin_key_addressing_grouped_fwd) -> user_o [B,D]."""
        a, D, P = self.args, self.dim, self.p_hop
        n_o = P + 1 if a.PS_O_ft else P
        o_cat = torch.empty((item.shape[0], n_o * D), dtype=torch.float32, device=self.device)
        w_h = self.h_emb_item_mlp_matrix.view(-1) if a.PS_O_ft else None
        groups = ops.group_pairs_by_user(user, n_user=uts.shape[0])
        rec = self.user_records(uts)
        if self._ka_flash_for(uts, rec, item.shape[0]):
            kp = getattr(self, "_ka_profile", None)      # bench.py: HIP events around the table build and the kernel
            if kp is not None:
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                evs[0].record()
            tabs = ops.key_addressing_flash_prepare(self.entity_emb_matrix, self.relation_emb_KGE_matrix, w_h, self.user_mlp_matrix, P,
                                                    out=self._ka_flash_tables(torch.cuda.current_stream().cuda_stream))
            if kp is not None:
                evs[1].record()
            uo = ops.key_addressing_flash(self.entity_emb_matrix, tabs, rec, groups, item, P, self.n_memory, self.n_relation,
                                          w_h is not None, self.user_mlp_bias, uts.shape[0])
            if kp is not None:
                evs[2].record()
                kp.append(evs)
            return uo
        if not ops.user_records_supported(self.dim, P, self.n_memory, self.n_relation, self.entity_emb_matrix.dtype == torch.bfloat16):
            rec = None                                   # (records built for the flash form's sake on a shape the records kernel does not take)
        er = ops.project_relations(self.entity_emb_matrix, self.relation_emb_KGE_matrix, w_h) if self._ka_er_for(uts, rec) else None
        ops.key_addressing_grouped(self.entity_emb_matrix, self.relation_emb_KGE_matrix, w_h, uts, groups, item, P,
                                   o_cat, n_o * D, self.n_relation, records=rec, er=er)
        return ops.linear([o_cat], self.user_mlp_matrix, D, bias=self.user_mlp_bias)

    def _forward_small(self, users, items, mem_h, mem_r, mem_t, uts):
        """The short way into the single-launch pass (the checks of forward_device that matter for it, nothing else): None
        when the call is not its case."""
        if not self._small_ok(items, mem_h, False):
            return None
        P, B = self.p_hop, items.shape[0]
        if uts is not None:
            self._check_uts(uts)
            if (uts.dtype == torch.int32 and uts.is_contiguous() and uts.shape[1] == P and uts.shape[3] == self.n_memory
                    and users.dtype == torch.int64 and users.dim() == 1 and users.shape[0] == B and users.is_contiguous()):
                return self._score_small_native(items, None, None, None, uts=uts, users=users)
            return None
        if mem_h is None or len(mem_h) < P or len(mem_r) < P or len(mem_t) < P:
            return None
        Nm = self.n_memory
        for lst in (mem_h, mem_r, mem_t):
            for m_ in lst[:P]:
                if m_.dtype != torch.int32 or m_.dim() != 2 or m_.shape[0] != B or m_.shape[1] != Nm or not m_.is_contiguous():
                    return None
        return self._score_small_native(items, mem_h, mem_r, mem_t)

    def forward_device(self, user_indices, item_indices, memories_h, memories_r, memories_t,
                       want_probs=False, uts=None, distinct_users=None):
        """model.py:125-159 on device-resident inputs (int64/int32 ids [B]; int32 ripple sets
        [B, n_memory] per hop).  Returns a namespace of device tensors.
        Shared-user form: ripple sets given as ONE [n_memory] list per hop (and ``user_indices`` a
        single id or [B]) score one user against ``item_indices`` -- see ``_key_addressing_shared``.
        ``uts`` (instead of the memories): see ``forward_users``."""
        a = self.args
        if not item_indices.is_cuda:
            raise RuntimeError("forward_device needs device-resident inputs (no CPU path)")
        if (self.small_max_batch and not want_probs and not self.validate_device_ids and item_indices.dim() == 1
                and item_indices.shape[0] <= self.small_max_batch and item_indices.is_contiguous()):
            out = self._forward_small(user_indices, item_indices, memories_h, memories_r, memories_t, uts)
            if out is not None:
                return out
        self._distinct_hint = distinct_users
        item32 = item_indices.contiguous()   # int64 (reference dtype) or int32: kernels take both
        user32 = user_indices.contiguous()
        need_ps = a.PS_only or (not a.HO_only) or a.User_orient_kg_eh
        grouped = False
        if uts is not None:
            self._check_uts(uts)
        if self.validate_device_ids:
            self._check_device_ids(item32, self.n_entity, "item_indices")
            self._check_device_ids(user32, uts.shape[0] if uts is not None else self.n_user, "user_indices")
            for lst, lim, what in ((memories_h, self.n_entity, "memories_h"), (memories_r, self.n_relation, "memories_r"),
                                   (memories_t, self.n_entity, "memories_t")):
                for m_ in (lst or ()):
                    self._check_device_ids(m_, lim, what)
        if uts is not None:
            if user32.numel() == 1 and item32.shape[0] > 1:
                user32 = user32.reshape(1).expand(item32.shape[0]).contiguous()
            # grouping pays when users repeat inside the batch (a user's rows are staged once per segment); with
            # about one pair per user the per-pair kernel is the faster one (measured: amazon-shaped, 32 768 pairs
            # of 70 585 users: 15.5 M vs 10.1 M pairs/s).  Static rule, no device sync: pairs per user >= 4.
            grouped = (need_ps and self.fused is not False and uts.dtype == torch.int32 and uts.is_contiguous()
                       and item32.shape[0] >= self.group_min_pairs_per_user * min(uts.shape[0], int(distinct_users or uts.shape[0]))
                       and ops.key_addressing_grouped_supported(self.dim, self.p_hop, self.n_memory, self.n_relation))
            feed_ok = (need_ps and uts.dtype == torch.int32 and uts.is_contiguous() and uts.dim() == 4
                       and uts.shape[1] == self.p_hop and uts.shape[3] == self.n_memory and user32.dtype == torch.int64
                       and user32.shape[0] == item32.shape[0])
            if feed_ok and self._small_ok(item32, None, want_probs):
                return self._score_small_native(item32, None, None, None, uts=uts, users=user32)     # ONE launch
            if (feed_ok and self._native_l2_ok(item32, None, want_probs, cap=not grouped)):
                # one native call; key addressing indexes user_triplet_set by users[b] itself, or -- grouped -- sorts the
                # batch by user on the device first (any batch size: the host issues ONE call per step instead of seven)
                return self._score_l2_native(item32, None, None, None, uts=uts, users=user32, grouped=grouped)
            users_kernel = (need_ps and not grouped and self.fused is not False and uts.dtype == torch.int32 and uts.is_contiguous()
                            and uts.dim() == 4 and uts.shape[1] == max(1, self.p_hop) and uts.shape[3] == self.n_memory
                            and user32.shape[0] == item32.shape[0] and user32.dtype in (torch.int64, torch.int32)
                            and ops.key_addressing_supported(self.n_memory, self.dim))
            if need_ps and not grouped and not users_kernel:     # any other wiring: assemble the per-pair feeds on the device
                sel = uts[user32.long()]
                P_ = sel.shape[1]
                memories_h = [sel[:, i, 0].contiguous() for i in range(P_)]
                memories_r = [sel[:, i, 1].contiguous() for i in range(P_)]
                memories_t = [sel[:, i, 2].contiguous() for i in range(P_)]
        else:
            users_kernel = False
        shared = (not grouped) and memories_h is not None and memories_h[0].dim() == 1   # one user's sets for the batch
        if not grouped and not shared and memories_h is not None \
                and all(m_.dtype == torch.int32 and m_.is_contiguous() for lst in (memories_h, memories_r, memories_t)
                        for m_ in lst[:self.p_hop]):
            if self._small_ok(item32, memories_h, want_probs):
                return self._score_small_native(item32, memories_h, memories_r, memories_t)          # ONE launch
            if self._native_l2_ok(item32, memories_h, want_probs):
                return self._score_l2_native(item32, memories_h, memories_r, memories_t)
        if shared and user32.numel() == 1:
            user32 = user32.reshape(1).expand(item32.shape[0]).contiguous()
        if shared and (self.n_memory % 4 != 0 or self.n_memory > 256):
            # outside the dense form (the [B, Nm] logits go through mvin_linear_fwd): replicate the sets
            rep = lambda lst: [m_.reshape(1, -1).expand(item32.shape[0], -1).contiguous() for m_ in lst]
            memories_h, memories_r, memories_t = rep(memories_h), rep(memories_r), rep(memories_t)
            shared = False
        if not need_ps:
            ps = None
        elif grouped:
            ps = self._key_addressing_grouped(user32, item32, uts)
        elif users_kernel:
            # the Python schedule of the users feed (batches above native_l2_max_batch, sharded tables): key addressing reads the
            # lists of users[b] out of user_triplet_set itself -- it was a torch gather + three copies per step to assemble them
            ps = self._key_addressing(user32, item32, None, None, None, uts=uts)
        elif shared:
            ps = self._key_addressing_shared(item32, [m_.contiguous() for m_ in memories_h],
                                             [m_.contiguous() for m_ in memories_r],
                                             [m_.contiguous() for m_ in memories_t])
        else:
            ps = self._key_addressing(user32, item32, memories_h, memories_r, memories_t)
        importance = []
        if a.PS_only:  # :142-144
            user_o = ps
            item_emb, scores, sig = ops.linear([self.entity_emb_matrix], None, self.dim, ids=[item32],
                                               score_u=user_o)
        else:
            user_o = self._lookup(self.user_emb_matrix, user32) if a.HO_only else ps
            if a.User_orient_kg_eh:
                q = ps
            else:
                q = user_o if a.HO_only else self._lookup(self.user_emb_matrix, user32)
            item_emb, scores, sig, importance = self.agg_fun(item32, q, user_o, want_probs)
        return SimpleNamespace(scores=scores, scores_normalized=sig, user_o=user_o,
                               item_embeddings=item_emb, importance_list=importance)

    # ------------------------------------------------------------------ feed handling
    def _check_uts(self, uts):
        """A device-resident user_triplet_set is validated ONCE per tensor (identity + torch version): the grouped /
        users-feed kernels index the entity table, the relation matrices and LDS with its entries."""
        # identity of the tensor OBJECT (a weak reference: a freed tensor's address is reused by the caching allocator) and its
        # torch version counter; the kernels clamp every id they index memory with, so this check is about raising the
        # reference's IndexError, not about memory safety
        ok = self._uts_ok
        if ok is not None and ok[0]() is uts and ok[1] == uts._version:
            return
        if uts.dim() != 4 or uts.shape[2] != 3 or uts.shape[1] != max(1, self.p_hop) or uts.shape[3] != self.n_memory:
            raise ValueError(f"user_triplet_set must be [n_user, {max(1, self.p_hop)}, 3, {self.n_memory}], "
                             f"got {tuple(uts.shape)}")
        P = self.p_hop
        bad = (uts[:, :, 0] < 0).any() | (uts[:, :, 0] >= self.n_entity).any()
        if P > 0:
            r, t = uts[:, :P, 1], uts[:, :P, 2]
            bad = bad | (r < 0).any() | (r >= self.n_relation).any() | (t < 0).any() | (t >= self.n_entity).any()
        if bool(bad.item()):
            raise IndexError("user_triplet_set: id out of range (heads / tails in [0, n_entity), relations in [0, n_relation))")
        self._uts_ok = (weakref.ref(uts), uts._version)

    def _check_device_ids(self, t, limit, what):
        if t is not None and t.numel() and bool(((t < 0) | (t >= limit)).any().item()):
            raise IndexError(f"{what}: id out of range [0, {limit})")

    def _to_device_ids(self, v, dtype, limit, what):
        if isinstance(v, torch.Tensor):
            t = v
            if not t.is_cuda:
                t = t.to(self.device)
            return t.to(dtype).contiguous()
        arr = np.asarray(v)
        if arr.size and (arr.min() < 0 or arr.max() >= limit):
            # the reference's CPU tf.gather raises InvalidArgument on out-of-range ids
            raise IndexError(f"{what}: id out of range [0, {limit})")
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.device).to(dtype).contiguous()

    def _feed(self, feed_dict):
        n_lists = max(1, self.p_hop)
        user = self._to_device_ids(feed_dict[self.user_indices], torch.int64, self.n_user, "user_indices")
        item = self._to_device_ids(feed_dict[self.item_indices], torch.int64, self.n_entity, "item_indices")
        mh = [self._to_device_ids(feed_dict[self.memories_h[i]], torch.int32, self.n_entity, "memories_h")
              for i in range(n_lists)]
        mr = [self._to_device_ids(feed_dict[self.memories_r[i]], torch.int32, self.n_relation, "memories_r")
              for i in range(n_lists)]
        mt = [self._to_device_ids(feed_dict[self.memories_t[i]], torch.int32, self.n_entity, "memories_t")
              for i in range(n_lists)]
        return user, item, mh, mr, mt

    # ------------------------------------------------------------------ run wrappers
    def train(self, sess, feed_dict):
        """model.py:416-417: one optimisation step (forward, loss of :378-412, backward, Adam of :414)
        on the GPU -> (None, loss).  The optimizer state lives in ``self.trainer``."""
        if self.trainer is None:
            from .training import Trainer
            self.trainer = Trainer(self)
        user, item, mh, mr, mt = self._feed(feed_dict)
        labels = torch.as_tensor(np.asarray(feed_dict[self.labels], dtype=np.float32)).to(self.device)
        # The reference trains at ONE batch size (train.py:56-64: `start += args.batch_size`, one sess.run per batch) and a step
        # at 512 / 1 024 pairs is ~60 short launches: from the second step of a batch size on, the step is one hipGraph replay
        # (training.GraphedTrainer: 0.68 vs 1.00 ms at 512 pairs).  MVIN_TRAIN_GRAPH=0, a multi-rank trainer or a batch above
        # train_graph_max_batch keep the eager launches; the first step of a size runs eagerly (it is the capture's warm-up).
        B = int(item.shape[0])
        if (os.environ.get("MVIN_TRAIN_GRAPH", "1") != "0" and self.trainer.world == 1 and B <= self.train_graph_max_batch):
            from .training import GraphedTrainer
            g = self._train_graphs.get(B)
            if g is False:
                g = None
            elif g is not None and g._storage_key() != g._captured:
                g = None                                   # adjacency / a parameter tensor replaced: capture again
            if g is None and self._train_seen == B and self._train_graphs.get(B) is not False:
                try:
                    g = self._train_graphs[B] = GraphedTrainer(self.trainer, B, ids_dtype=item.dtype, warmup=0)
                except RuntimeError:
                    # a step that cannot be captured (a kernel or allocation that refuses to run under capture) still runs
                    # eagerly: remember the failure for this batch size and fall through (train_epoch_device does the same)
                    g, self._train_graphs[B] = None, False
                if len(self._train_graphs) > 4:
                    self._train_graphs.pop(next(iter(self._train_graphs)))
            self._train_seen = B
            if g:
                return None, float(g.step(user, item, labels, mh, mr, mt).item())
        return None, self.trainer.step(user, item, labels, mh, mr, mt)

    def get_scores(self, sess, feed_dict):
        """model.py:443-444."""
        user, item, mh, mr, mt = self._feed(feed_dict)
        out = self.forward_device(user, item, mh, mr, mt)
        return item.cpu().numpy(), out.scores_normalized.cpu().numpy()

    def eval(self, sess, feed_dict):
        """model.py:419-426."""
        from sklearn.metrics import f1_score, roc_auc_score
        user, item, mh, mr, mt = self._feed(feed_dict)
        labels = np.asarray(feed_dict[self.labels], dtype=np.float32)
        scores = self.forward_device(user, item, mh, mr, mt).scores_normalized.cpu().numpy()
        auc = roc_auc_score(y_true=labels, y_score=scores)
        scores[scores >= 0.5] = 1
        scores[scores < 0.5] = 0
        f1 = f1_score(y_true=labels, y_pred=scores)
        acc = np.mean(np.equal(scores, labels))
        return auc, acc, f1

    def eval_case_study(self, sess, feed_dict):
        """model.py:428-441: ids of every level plus the attention weights of the i=0 pass."""
        user, item, mh, mr, mt = self._feed(feed_dict)
        out = self.forward_device(user, item, mh, mr, mt, want_probs=True)
        L = self.n_mix_hop * self.h_hop
        ents, rels = self.get_neighbors(item.to(torch.int32), levels=L)
        imp = out.importance_list
        imp0 = imp[0].cpu().numpy() if imp and imp[0] is not None else None
        imp1 = imp[1].cpu().numpy() if len(imp) > 1 and imp[1] is not None else 0
        return (user.cpu().numpy(), np.asarray(feed_dict[self.labels], dtype=np.float32), item.cpu().numpy(),   # fetched placeholder: float32 (model.py:52)
                [e.cpu().numpy().astype(np.int64) for e in ents],
                [r.cpu().numpy().astype(np.int64) for r in rels], imp0, imp1)
