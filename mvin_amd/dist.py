"""Multi-GPU layer: one process per GPU, pairs split across ranks, the entity embedding
table row-sharded across the ranks of a node, cross-shard rows fetched with all-to-all
(RCCL over xGMI when the process group backend is "nccl").

The reference has no distributed code at all; this is new design
mandated by BASELINE.json's north_star.  Layout:

  * cyclic row ownership, owner(x) = x mod W (item ids occupy [0, n_item) and the
    Zipf-hot rows are items, so contiguous blocks would put them all on the first ranks).  Rank r
    stores rows r, r+W, r+2W, ... as one contiguous shard of n_local = ceil(nE / W) rows (zero padded).
  * SHARD SPACE: the sharded model works on relabelled entity ids  pi(x) = (x mod W) * n_local + x div W,
    in which rank r's shard is the contiguous block [r*n_local, (r+1)*n_local).  The adjacency tables
    are relabelled once at construction (rows permuted, entries mapped), a device-resident
    user_triplet_set once, and a batch's item ids (and per-pair ripple sets, if fed that way) by one
    elementwise op per step.  Every kernel then runs unchanged on a WORKING TABLE of W*n_local rows
    addressed by shard-space id, and an all-gather of the shards IS that table (no transpose pass).
  * adjacency (int32), the relation table, the KGE relation matrices, the user table and all dense
    weights are replicated (a few MB);
  * every rank scores its own slice of the batch (pairs are independent: no reduction);
  * every step a rank brings the rows its pairs touch into the working table, in one of two regimes:

    dense  (batch footprint >= table: B*K^L row references vs nE rows -- every benchmark config):
            every rank receives every shard.  Over RCCL this is ONE grouped send/recv
            (torch.distributed.all_to_all on W views of the SAME local shard -- no W-fold send
            staging -- and W slices of the working table): direct P2P on all 7 xGMI links at once,
            no ring, static sizes, no host sync.  Other backends (gloo in the CPU tests):
            all_gather_into_tensor, same result.
    sparse (small batches / tables much larger than a batch's footprint): the touched rows are marked
            through the replicated adjacency (tree levels 0..L + ripple-set heads and tails,
            deduplicated level by level), then all-to-all (counts) -> all-to-all-v (ids) -> owner-side
            HIP row gather (mvin_gather_rows, fp32 or bf16 rows) -> all-to-all-v (rows) -> HIP scatter
            into the working table.

Memory per rank: the shard (1/W of the table) + one working table (two with the pipeline, so that
the exchange for step i+1 runs on a side stream while step i is scored).  The working table is
scratch: it is refilled every step (the exchange is never skipped or cached across steps).

The regime must be the SAME on every rank (the two regimes issue different collectives): it is a
function of the GLOBAL batch size only (``global_batch`` argument, default local batch * W, i.e. equal
slices), never of a rank's own slice; MVIN_DIST_CHECK=1 verifies the agreement with an all-reduce.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def n_local_rows(n_entity, world):
    return -(-n_entity // world)


def to_shard_space(ids, n_entity, world):
    """pi(x) = (x mod W) * n_local + x div W (torch tensor or numpy array, any integer dtype)."""
    if world == 1:
        return ids
    nl = n_local_rows(n_entity, world)
    return (ids % world) * nl + ids // world


def from_shard_space(pids, n_entity, world):
    if world == 1:
        return pids
    nl = n_local_rows(n_entity, world)
    return (pids % nl) * world + pids // nl


def shard_rows(table, rank, world):
    """Rows owned by ``rank`` (x mod W == rank, ascending), zero-padded to n_local rows."""
    nl = n_local_rows(table.shape[0], world)
    mine = table[rank::world]
    out = torch.zeros((nl, table.shape[1]), dtype=table.dtype, device=table.device)
    out[:mine.shape[0]] = mine
    return out


def permute_adjacency(adj_entity, adj_relation, n_entity, world):
    """The fixed-fan-out adjacency in shard space: row pi(x) holds pi(adj_entity[x]) / adj_relation[x];
    padding rows are zero (= entity 0 / relation 0, like entities absent from the KG,
    data_loader_user_set.py:377-380).  numpy in, numpy int32 out."""
    adj_entity, adj_relation = np.asarray(adj_entity), np.asarray(adj_relation)
    nl = n_local_rows(n_entity, world)
    K = adj_entity.shape[1]
    pe = np.zeros((world * nl, K), dtype=np.int32)
    pr = np.zeros((world * nl, K), dtype=np.int32)
    pos = to_shard_space(np.arange(n_entity, dtype=np.int64), n_entity, world)
    pe[pos] = to_shard_space(adj_entity[:n_entity].astype(np.int64), n_entity, world)
    pr[pos] = adj_relation[:n_entity]
    return pe, pr


def permute_ripple_sets(uts, n_entity, world):
    """user_triplet_set [n_user, P, 3, Nm] with heads and tails relabelled (relations untouched)."""
    out = uts.clone() if torch.is_tensor(uts) else np.array(uts, copy=True)
    out[:, :, 0] = to_shard_space(uts[:, :, 0], n_entity, world)
    out[:, :, 2] = to_shard_space(uts[:, :, 2], n_entity, world)
    return out


def mark_needed(n_rows, adj_entity, items, levels, extra_ids=()):
    """Boolean mask [n_rows] of the entity rows a batch touches: the K-ary tree below each item down
    to ``levels`` (model.py:243-256, through the replicated adjacency) plus ``extra_ids``
    (ripple-set heads and tails).  Level sets are deduplicated level by level, so the work is
    bounded by nE*K per level instead of B*K^L.  Ids and adjacency in the same id space."""
    dev = adj_entity.device
    need = torch.zeros(n_rows, dtype=torch.bool, device=dev)
    frontier = torch.zeros(n_rows, dtype=torch.bool, device=dev)
    frontier[items.long()] = True
    need |= frontier
    for _ in range(levels):
        cur = frontier.nonzero(as_tuple=True)[0]
        frontier = torch.zeros(n_rows, dtype=torch.bool, device=dev)
        frontier[adj_entity[cur].reshape(-1).long()] = True
        need |= frontier
    for ids in extra_ids:
        need[ids.reshape(-1).long()] = True
    return need


def mark_needed_static(n_rows, adj_entity, items, levels, extra_ids=()):
    """mark_needed without a host sync: no ``nonzero`` (whose result size the host must read) -- every level pushes the
    frontier through ALL adjacency rows with one index_add over n_rows * K entries (3.6 M at C3: tens of microseconds)."""
    dev = adj_entity.device
    K = adj_entity.shape[1]
    flat = adj_entity.reshape(-1).long()
    frontier = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    frontier[items.long()] = 1
    need = frontier.clone()
    for _ in range(levels):
        nxt = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        nxt.index_add_(0, flat, (frontier > 0).to(torch.int32).repeat_interleave(K))
        frontier = nxt
        need += nxt
    for ids in extra_ids:
        need[ids.reshape(-1).long()] = 1
    return need > 0


def hip_row_gather(table, idx_int32):
    """Owner-side row gather on the GPU (mvin_gather_rows: fp32 or bf16 rows, moved untouched)."""
    from . import ops
    return ops.gather_rows(table, idx_int32.contiguous())


def hip_row_scatter(table, idx_int32, rows):
    from . import ops
    return ops.scatter_rows(table, idx_int32.contiguous(), rows.contiguous())


def _torch_row_scatter(table, idx_int32, rows):
    table.index_copy_(0, idx_int32.long(), rows)
    return table


class ShardedEntityTable(object):
    """Row-sharded entity table + per-step row fetch into a working table (all ids in shard space)."""

    def __init__(self, local_rows, n_entity, rank, world, row_gather=None, row_scatter=None, group=None,
                 always_collective=False):
        """``local_rows``: this rank's padded shard ([n_local, D], see shard_rows).
        ``row_gather(table, idx_int32) -> rows`` / ``row_scatter(table, idx_int32, rows)``: the row movers of
        the sparse regime (the HIP kernels by default; the CPU tests inject torch indexing)."""
        self.n_entity, self.rank, self.world = n_entity, rank, world
        self.n_local = n_local_rows(n_entity, world)
        self.lo = rank * self.n_local
        if local_rows.shape[0] != self.n_local:
            raise ValueError(f"shard must have n_local={self.n_local} rows (padded), got {local_rows.shape[0]}")
        self.local = local_rows.contiguous()
        self.group = group
        self.always_collective = always_collective   # run the collectives even when world == 1 (tests)
        on_gpu = self.local.is_cuda
        self.row_gather = row_gather or (hip_row_gather if on_gpu else (lambda t, i: t[i.long()]))
        self.row_scatter = row_scatter or (hip_row_scatter if on_gpu else _torch_row_scatter)
        self.dim = local_rows.shape[1]
        self.work = self.new_work_table()
        self.refreshes = 0
        self.last_stats = {}

    def refresh(self):
        """Call after the shard's rows changed in place through raw pointers (nothing is staged; tables derived
        from the working table are rebuilt after the next exchange)."""
        self.refreshes += 1

    def new_work_table(self):
        """A working table: W*n_local rows (>= nE), addressed by shard-space id."""
        return torch.zeros((self.world * self.n_local, self.dim), dtype=self.local.dtype, device=self.local.device)

    def bytes_per_rank(self, n_work_tables=1):
        row = self.dim * self.local.element_size()
        return {"shard": self.n_local * row, "working_tables": n_work_tables * self.world * self.n_local * row}

    # ---- dense regime ---------------------------------------------------------------------
    def fetch_all(self, work=None):
        """Every rank receives every shard; shard r lands at rows [r*n_local, (r+1)*n_local)."""
        work = self.work if work is None else work
        W = self.world
        if W == 1 and not self.always_collective:
            work.copy_(self.local)
        elif dist.get_backend(self.group) == "nccl":
            # grouped P2P (ncclGroupStart; W x send/recv; ncclGroupEnd): every peer pair uses its own xGMI link.
            # The W inputs are the SAME tensor -- nothing is replicated on the sender.
            dist.all_to_all(list(work.view(W, self.n_local, self.dim).unbind(0)), [self.local] * W, group=self.group)
        else:
            dist.all_gather_into_tensor(work, self.local, group=self.group)
        self.last_stats = {"mode": "dense", "requested": self.n_entity,
                           "remote": self.n_entity - len(range(self.rank, self.n_entity, W))}
        return work

    # ---- sparse regime --------------------------------------------------------------------
    def fetch(self, need_mask, work=None):
        """Make ``work[p]`` valid for every shard-space id p with need_mask[p] (default: ``self.work``)."""
        work = self.work if work is None else work
        W = self.world
        dev = self.local.device
        ids = need_mask.nonzero(as_tuple=True)[0]            # sorted: already grouped by owner block
        send_counts = torch.bincount(ids // self.n_local, minlength=W)
        if W == 1 and not self.always_collective:
            i32 = ids.to(torch.int32)
            self.row_scatter(work, i32, self.row_gather(self.local, i32))
            self.last_stats = {"mode": "sparse", "requested": int(ids.numel()), "remote": 0}
            return work
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()   # one host sync per step
        want = torch.empty(sum(rc), dtype=ids.dtype, device=dev)     # ids other ranks want from me
        dist.all_to_all_single(want, ids, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        out_rows = self.row_gather(self.local, (want - self.lo).to(torch.int32))
        got = torch.empty((ids.numel(), self.dim), dtype=self.local.dtype, device=dev)
        dist.all_to_all_single(got, out_rows, output_split_sizes=sc, input_split_sizes=rc, group=self.group)
        self.row_scatter(work, ids.to(torch.int32), got)
        self.last_stats = {"mode": "sparse", "requested": int(ids.numel()),
                           "remote": int(ids.numel()) - sc[self.rank], "served": int(want.numel())}
        return work


class _LazyStats(dict):
    """Exchange statistics whose counts stay on the device until somebody reads them (reading is the host sync)."""
    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return int(v) if torch.is_tensor(v) else v

    def get(self, k, default=None):
        return self[k] if k in self else default


def _fetch_static(self, need_mask, capacity, work=None):
    """The sparse regime WITHOUT a host sync: fixed-capacity id buffers.  Every rank asks every owner for exactly
    ``C = min(capacity, n_local)`` rows -- the needed local ids of that owner first, the rest repeating its local row 0
    (fetched and written again with its own content: harmless) -- so both collectives have static, equal splits: no count
    exchange, no ``nonzero`` / ``.tolist()``, two collectives instead of three, and the step can be captured in a graph.
    ``capacity`` must bound the distinct rows a rank can need from one owner (ShardedMVIN passes the row references of
    the rank's batch share, a static number); ``self.overflow`` (device flag, never read here) is set when it did not."""
    work = self.work if work is None else work
    W, nl, dev = self.world, self.n_local, self.local.device
    C = int(min(capacity, nl))
    m = need_mask.view(W, nl)
    order = torch.sort(m.to(torch.uint8), dim=1, descending=True, stable=True).indices[:, :C]      # needed local ids first
    cnt = m.sum(1)
    valid = torch.arange(C, device=dev)[None, :] < cnt[:, None]
    ids_local = torch.where(valid, order, torch.zeros_like(order)).to(torch.int32).contiguous()     # [W, C]
    self.overflow = (cnt > C).any() if getattr(self, "overflow", None) is None else (self.overflow | (cnt > C).any())
    if W == 1 and not self.always_collective:
        self.row_scatter(work, ids_local.view(-1), self.row_gather(self.local, ids_local.view(-1)))
    else:
        want = torch.empty_like(ids_local)                   # row w: the local ids rank w wants from me
        dist.all_to_all_single(want, ids_local, group=self.group)
        out_rows = self.row_gather(self.local, want.view(-1))
        got = torch.empty((W * C, self.dim), dtype=self.local.dtype, device=dev)
        dist.all_to_all_single(got, out_rows, group=self.group)
        dest = (ids_local + (torch.arange(W, device=dev, dtype=torch.int32) * nl)[:, None]).view(-1)
        self.row_scatter(work, dest, got)
    row = self.dim * self.local.element_size()
    self.last_stats = _LazyStats({"mode": "sparse", "static": True, "capacity_per_owner": C, "requested": cnt.sum(),
                                  "remote": cnt.sum() - cnt[self.rank], "wire_bytes_per_rank": (W - 1) * C * (4 + row)})
    return work


ShardedEntityTable.fetch_static = _fetch_static


def exchange_wire_bytes(n_entity, row_bytes, world, local_pairs, K, depth, nodes_level=None):
    """Bytes a rank RECEIVES per step, by exchange design (the measured argument of DESIGN.md section 5):
      replicate   : every shard to every rank (dense regime): (W-1)/W of the table;
      partial_sums: SURVEY 8(e)'s owner-side partial sums -- one row-sized vector per requesting node of level
                    depth-1 and per REMOTE owner (+ nothing else: owners recompute the weights from the replicated
                    adjacency): local_pairs * K^(depth-1) * (W-1) * row bytes.
    Partial sums win only while a rank's level-(depth-1) nodes are fewer than n_entity / W -- i.e. for batches far
    smaller than any BASELINE config (C3 at 65 536 pairs per rank: 2.1 M nodes vs 13 k rows per shard)."""
    nl = n_local_rows(n_entity, world)
    nodes = local_pairs * K ** max(depth - 1, 0) if nodes_level is None else nodes_level
    return {"replicate": (world - 1) * nl * row_bytes, "partial_sums": nodes * (world - 1) * row_bytes}


class ShardedMVIN(object):
    """MVIN scoring with the entity table row-sharded over the ranks of ``group``.

    Build with ``ShardedMVIN.build(...)``: it creates an mvin_amd.model.MVIN that lives in shard space
    (relabelled adjacency, W*n_local entity rows) and whose ``entity_emb_matrix`` is the working table
    of a ShardedEntityTable; every forward first brings in the rows the local pairs need.  Callers keep
    passing ORIGINAL entity ids (items, ripple sets, user_triplet_set)."""

    def __init__(self, model, shard, n_entity, rank, world, group=None, row_gather=None, row_scatter=None,
                 always_collective=False, regime="auto"):
        self.model, self.rank, self.world, self.n_entity = model, rank, world, n_entity
        self.table = ShardedEntityTable(shard.to(model.device), n_entity, rank, world, row_gather, row_scatter,
                                        group, always_collective)
        if model.n_entity != self.table.work.shape[0]:
            raise ValueError("the model must be built in shard space: n_entity = W * n_local rows (ShardedMVIN.build)")
        self.regime = regime
        self.group = group
        model.entity_emb_matrix = self.table.work
        self._bufs = None
        self._uts = None
        self._check = os.environ.get("MVIN_DIST_CHECK") == "1"
        self._exchanges = 0
        self._buf_token = [None, None]
        self._item_cache = None
        # sparse regime: fixed-capacity id buffers (no host sync); MVIN_DIST_DYNAMIC=1 keeps the count-exchange form
        self.static_sparse = os.environ.get("MVIN_DIST_DYNAMIC") != "1"

    @classmethod
    def build(cls, args, n_user, n_entity, n_relation, adj_entity, adj_relation, params, shard, rank, world,
              device=None, table_dtype="f32", **kw):
        """``params``: the replicated parameters (an ``entity_emb_matrix`` entry is ignored);
        ``shard``: this rank's rows (shard_rows(full_table, rank, world)), fp32 or bf16."""
        from .model import MVIN
        mkw = {k: kw.pop(k) for k in ("hoist", "fused", "seed") if k in kw}
        pe, pr = permute_adjacency(adj_entity, adj_relation, n_entity, world)
        rows = pe.shape[0]
        p = dict(params, entity_emb_matrix=np.zeros((rows, shard.shape[1]), dtype=np.float32))
        model = MVIN(args, n_user, rows, n_relation, pe, pr, params=p, device=device, table_dtype=table_dtype, **mkw)
        if shard.dtype != model.entity_emb_matrix.dtype:
            shard = shard.to(model.entity_emb_matrix.dtype)
        return cls(model, shard, n_entity, rank, world, **kw)

    # ---- id spaces ------------------------------------------------------------------------
    def ids(self, x):
        """Original entity ids -> shard space.  Device tensors take one HIP launch (mvin_shard_space_ids; the
        torch spelling is four elementwise kernels per call, ~25 us of a rank's 1.2 ms step at W = 8)."""
        if torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.int64, torch.int32) and \
                (self.world > 1 or self.table.always_collective):
            from . import ops
            return ops.shard_space_ids(x, self.world, self.table.n_local)
        return to_shard_space(x, self.n_entity, self.world)

    def set_user_triplet_set(self, uts):
        """Register a device-resident user_triplet_set (original ids): relabelled once, then
        ``forward_users`` needs no per-step id work beyond the item ids."""
        self._uts = permute_ripple_sets(uts.to(self.model.device), self.n_entity, self.world).contiguous()
        return self._uts

    def _depth(self):
        m = self.model
        if m.args.PS_only:
            return 0
        return m.n_mix_hop * m.h_hop if m.args.wide_deep else m.h_hop

    def is_dense(self, global_batch):
        """Regime from the GLOBAL batch only (identical on every rank; no device sync): row references of the
        batch's per-rank share vs table rows.  The static sparse form asks every owner for min(refs, n_local) rows, so from
        refs >= n_local on it would move the whole table AND pay the id exchange, the sort and the scatter on top: that band
        is dense too.  The count-exchange form (MVIN_DIST_DYNAMIC=1) moves distinct rows and keeps the table-size
        threshold."""
        if self.regime != "auto":
            return self.regime == "dense"
        m = self.model
        per_rank = -(-int(global_batch) // self.world)
        refs = per_rank * (sum(m.n_neighbor ** e for e in range(self._depth() + 1)) + 2 * m.n_memory * max(1, m.p_hop))
        return refs >= (self.table.n_local if self.static_sparse else self.n_entity)

    def _regime(self, local_batch, global_batch):
        dense = self.is_dense(local_batch * self.world if global_batch is None else global_batch)
        if self._check and (self.world > 1 or self.table.always_collective):
            t = torch.tensor([int(dense), -int(dense)], device=self.model.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            if int(t[0]) != -int(t[1]):
                raise RuntimeError("ranks chose different exchange regimes: pass the same global_batch everywhere")
        return dense

    def needed(self, item_p, heads_tails_p):
        m = self.model
        return mark_needed(self.table.work.shape[0], m.adj_entity, item_p, self._depth(), heads_tails_p)

    def _ripple_ids(self, users, mem_h_p, mem_t_p):
        m = self.model
        need_ps = m.args.PS_only or (not m.args.HO_only) or m.args.User_orient_kg_eh
        if not need_ps:
            return []
        if mem_h_p is not None:
            return list(mem_h_p[:max(1, m.p_hop)]) + list(mem_t_p[:m.p_hop])
        sel = self._uts[users.long()]
        return [sel[:, :, 0], sel[:, :m.p_hop, 2]]

    def _exchange(self, work, users, item_p, mem_h_p, mem_t_p, global_batch):
        """Fill ``work`` and return its CONTENT TOKEN.  The row movers and collectives write the working table
        through raw pointers, so torch's version counter does not see them: tables derived from the working table
        (entity-table mode, MVIN._hoist_key) are keyed on this token instead.  Dense regime: the content is the
        whole table, the same for every step until the shard changes (its version counter / refresh());
        sparse regime: a different row set every exchange, so a fresh token every time."""
        if self._regime(item_p.shape[0], global_batch):
            self.table.fetch_all(work)
            return ("dense", self.table.local._version, self.table.refreshes)
        m = self.model
        need = mark_needed_static(self.table.work.shape[0], m.adj_entity, item_p, self._depth(),
                                  self._ripple_ids(users, mem_h_p, mem_t_p))
        if self.static_sparse:
            # fixed-capacity buffers: no host sync anywhere in the step (capacity = the batch share's row references)
            refs = item_p.shape[0] * (sum(m.n_neighbor ** e for e in range(self._depth() + 1)) + 2 * m.n_memory * max(1, m.p_hop))
            self.table.fetch_static(need, refs, work)
            if self._check and bool(self.table.overflow):      # debug mode only: reading the flag is a host sync
                raise RuntimeError("static sparse exchange: a rank needed more rows from one owner than the capacity "
                                   f"{min(refs, self.table.n_local)} (rows were dropped)")
        else:
            self.table.fetch(need, work)
        self._exchanges += 1
        return ("sparse", self._exchanges)

    def _item_ids(self, item_indices):
        """Shard-space item ids, remembered for the tensor they came from (identity + torch version): the pipelined step maps
        the same batch twice -- prefetch() for the exchange, forward_prefetched() for the scoring -- and the relabelling launch
        is 8 us of a rank's 0.49 ms step at W = 8."""
        c = self._item_cache
        if c is not None and c[0] is item_indices and c[1] == item_indices._version:
            if torch.is_tensor(c[2]) and c[2].is_cuda:
                c[2].record_stream(torch.cuda.current_stream())      # made on the exchange stream, read on this one
            return c[2]
        item_p = self.ids(item_indices)
        self._item_cache = (item_indices, item_indices._version, item_p) if torch.is_tensor(item_indices) else None
        return item_p

    def _map_feed(self, item_indices, memories_h, memories_t):
        item_p = self._item_ids(item_indices)
        if memories_h is None:
            return item_p, None, None
        return item_p, [self.ids(t) for t in memories_h], [self.ids(t) for t in memories_t]

    def forward_device(self, user_indices, item_indices, memories_h, memories_r, memories_t, global_batch=None, **kw):
        """Exchange, then score (serialised on the current stream).  Original entity ids in."""
        item_p, mh_p, mt_p = self._map_feed(item_indices, memories_h, memories_t)
        self.model._table_token = self._exchange(self.table.work, user_indices, item_p, mh_p, mt_p, global_batch)
        self.model.entity_emb_matrix = self.table.work
        if memories_h is None:
            return self.model.forward_users(user_indices, item_p, self._uts, **kw)
        return self.model.forward_device(user_indices, item_p, mh_p, memories_r, mt_p, **kw)

    def forward_users(self, user_indices, item_indices, global_batch=None, **kw):
        """The same with the ripple sets taken from the registered user_triplet_set (set_user_triplet_set)."""
        if self._uts is None:
            raise RuntimeError("call set_user_triplet_set(uts) first")
        return self.forward_device(user_indices, item_indices, None, None, None, global_batch=global_batch, **kw)

    # ---- double-buffered pipeline: exchange for batch i+1 overlaps the scoring of batch i ----
    def enable_pipeline(self):
        self._bufs = [self.table.work, self.table.new_work_table()]
        self._side = torch.cuda.Stream(device=self.model.device, priority=-1)
        self._ready = [None, None]      # event: rows of buffer b are in place (side stream)
        self._free = [None, None]       # event: the scoring that read buffer b is done (main stream)

    def prefetch(self, buf, user_indices, item_indices, memories_h=None, memories_t=None, global_batch=None):
        """Start the row exchange for a batch into working table ``buf`` on the side stream."""
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._side):
            if self._free[buf] is None:
                self._side.wait_stream(main)
            else:
                self._side.wait_event(self._free[buf])
            item_p, mh_p, mt_p = self._map_feed(item_indices, memories_h, memories_t)
            self._buf_token[buf] = self._exchange(self._bufs[buf], user_indices, item_p, mh_p, mt_p, global_batch)
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._ready[buf] = ev

    def forward_prefetched(self, buf, user_indices, item_indices, memories_h, memories_r, memories_t, **kw):
        """Score a batch whose rows were prefetched into ``buf``."""
        main = torch.cuda.current_stream()
        main.wait_event(self._ready[buf])
        self.model.entity_emb_matrix = self._bufs[buf]
        self.model._table_token = self._buf_token[buf]
        item_p, mh_p, mt_p = self._map_feed(item_indices, memories_h, memories_t)
        if memories_h is None:
            out = self.model.forward_users(user_indices, item_p, self._uts, **kw)
        else:
            out = self.model.forward_device(user_indices, item_p, mh_p, memories_r, mt_p, **kw)
        done = torch.cuda.Event()
        done.record(main)
        self._free[buf] = done
        return out
