"""Multi-GPU layer: one process per GPU, pairs split across ranks, the entity embedding
table row-sharded across the ranks of a node, cross-shard rows fetched with all-to-all
(RCCL over xGMI when the process group backend is "nccl").

The reference has no distributed code at all; this is new design
mandated by BASELINE.json's north_star.  Layout:

  * contiguous row blocks: n_local = ceil(nE / world), owner(x) = x // n_local; every shard is
    padded to n_local rows so all exchanges have static sizes;
  * adjacency (int32, [nE,K]), the relation table, the KGE relation matrices, the user table
    and all dense weights are replicated (a few MB);
  * every rank scores its own slice of the batch (pairs are independent: no reduction);
  * every step a rank brings the rows its pairs touch into a WORKING TABLE addressed by global
    entity id (so the scoring kernels run unchanged), in one of two regimes:

    dense  (batch footprint >= table: B*K^L row references vs nE rows -- every benchmark
            config): ONE all_to_all_single in which each rank sends its shard to every peer
            (RCCL lowers it to grouped P2P send/recv over all 7 xGMI links at once, no ring),
            i.e. each row crosses the fabric once per rank per step, no id traffic, no host sync;
    sparse (small batches / tables much larger than a batch's footprint): the touched rows
            are marked through the replicated adjacency (tree levels 0..L + ripple-set heads and
            tails, deduplicated level by level), then all-to-all (counts) -> all-to-all-v (ids)
            -> owner-side HIP row gather -> all-to-all-v (rows) -> scatter into the working table.

The working table is full-size address space but scratch: it is refilled every step (the
exchange is never skipped or cached across steps), and a second working table lets the exchange
for step i+1 run on a side stream while step i is scored.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_entity, rank, world):
    n_local = -(-n_entity // world)
    lo = min(n_entity, rank * n_local)
    return lo, min(n_entity, lo + n_local), n_local


def shard_rows(table, rank, world):
    """Rows owned by ``rank`` (contiguous block partition), zero-padded to n_local rows."""
    lo, hi, n_local = shard_bounds(table.shape[0], rank, world)
    out = torch.zeros((n_local, table.shape[1]), dtype=table.dtype, device=table.device)
    out[:hi - lo] = table[lo:hi]
    return out


def mark_needed(n_entity, adj_entity, items, levels, extra_ids=()):
    """Boolean mask [nE] of the entity rows a batch touches: the K-ary tree below each item down
    to ``levels`` (model.py:243-256, through the replicated adjacency) plus ``extra_ids``
    (ripple-set heads and tails).  Level sets are deduplicated level by level, so the work is
    bounded by nE*K per level instead of B*K^L."""
    dev = adj_entity.device
    need = torch.zeros(n_entity, dtype=torch.bool, device=dev)
    frontier = torch.zeros(n_entity, dtype=torch.bool, device=dev)
    frontier[items.long()] = True
    need |= frontier
    for _ in range(levels):
        cur = frontier.nonzero(as_tuple=True)[0]
        frontier = torch.zeros(n_entity, dtype=torch.bool, device=dev)
        frontier[adj_entity[cur].reshape(-1).long()] = True
        need |= frontier
    for ids in extra_ids:
        need[ids.reshape(-1).long()] = True
    return need


class ShardedEntityTable(object):
    """Row-sharded entity table + per-step all-to-all row fetch into a working table."""

    def __init__(self, local_rows, n_entity, rank, world, row_gather, group=None, always_collective=False):
        """``local_rows``: this rank's padded shard ([n_local, D], see shard_rows).
        ``row_gather(table, idx_int32) -> rows``: the owner-side gather of the sparse regime
        (the HIP gather kernel in production: ``hip_row_gather``)."""
        self.n_entity, self.rank, self.world = n_entity, rank, world
        self.lo, self.hi, self.n_local = shard_bounds(n_entity, rank, world)
        if local_rows.shape[0] != self.n_local:
            raise ValueError(f"shard must have n_local={self.n_local} rows (padded), got {local_rows.shape[0]}")
        self.local = local_rows.contiguous()
        self.group = group
        self.always_collective = always_collective   # run the collectives even when world == 1 (tests)
        self.row_gather = row_gather
        self.dim = local_rows.shape[1]
        self.work = self.new_work_table()
        self._send = None
        self.last_stats = {}

    def refresh(self):
        """Call after the shard's rows changed in place (e.g. a training step)."""
        self._send = None

    def new_work_table(self):
        """A working table: world*n_local rows (>= nE), addressed by global entity id."""
        return torch.zeros((self.world * self.n_local, self.dim), dtype=self.local.dtype, device=self.local.device)

    # ---- dense regime ---------------------------------------------------------------------
    def fetch_all(self, work=None):
        """Every rank sends its shard to every peer in one all-to-all (direct P2P on all links)."""
        work = self.work if work is None else work
        W = self.world
        if W == 1 and not self.always_collective:
            work.copy_(self.local)
        else:
            if self._send is None:   # the shard laid out once per destination (rebuilt by refresh())
                self._send = self.local.unsqueeze(0).expand(W, self.n_local, self.dim).contiguous()
            dist.all_to_all_single(work, self._send.view(W * self.n_local, self.dim), group=self.group)
        self.last_stats = {"mode": "dense", "requested": self.n_entity,
                           "remote": self.n_entity - (self.hi - self.lo)}
        return work

    # ---- sparse regime --------------------------------------------------------------------
    def fetch(self, need_mask, work=None):
        """Make ``work[x]`` valid for every x with need_mask[x] (default: ``self.work``)."""
        work = self.work if work is None else work
        W = self.world
        dev = self.local.device
        ids = need_mask.nonzero(as_tuple=True)[0]            # sorted: already grouped by owner block
        send_counts = torch.bincount(ids // self.n_local, minlength=W)
        if W == 1 and not self.always_collective:
            work.index_copy_(0, ids, self.row_gather(self.local, ids.to(torch.int32)))
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
        work.index_copy_(0, ids, got)
        self.last_stats = {"mode": "sparse", "requested": int(ids.numel()),
                           "remote": int(ids.numel()) - sc[self.rank], "served": int(want.numel())}
        return work


def hip_row_gather(table, idx_int32):
    """Owner-side row gather on the GPU: mvin_linear_fwd in its identity/gather form."""
    from . import ops
    if table.dtype != torch.float32:   # bf16 shard: rows move as bf16 (pure data movement)
        return table.index_select(0, idx_int32.long())
    return ops.linear([table], None, table.shape[1], ids=[idx_int32.contiguous()])


class ShardedMVIN(object):
    """MVIN scoring with the entity table row-sharded over the ranks of ``group``.

    ``model`` is an mvin_amd.model.MVIN whose ``entity_emb_matrix`` is replaced by a working
    table of a ShardedEntityTable; every forward first brings in the rows the local pairs need."""

    def __init__(self, model, full_entity_table_or_shard, rank, world, group=None, row_gather=None,
                 is_shard=False, always_collective=False, regime="auto"):
        self.model, self.rank, self.world = model, rank, world
        local = full_entity_table_or_shard if is_shard else shard_rows(full_entity_table_or_shard, rank, world)
        self.table = ShardedEntityTable(local.to(model.device), model.n_entity, rank, world,
                                        row_gather or hip_row_gather, group, always_collective)
        self.regime = regime
        model.entity_emb_matrix = self.table.work
        self._bufs = None

    def _depth(self):
        m = self.model
        if m.args.PS_only:
            return 0
        return m.n_mix_hop * m.h_hop if m.args.wide_deep else m.h_hop

    def is_dense(self, batch):
        """Static regime choice (no device sync): row references of the batch vs table rows."""
        if self.regime != "auto":
            return self.regime == "dense"
        m = self.model
        refs = batch * (sum(m.n_neighbor ** e for e in range(self._depth() + 1)) + 2 * m.n_memory * max(1, m.p_hop))
        return refs >= m.n_entity

    def needed(self, item_indices, memories_h, memories_t):
        m = self.model
        extra = []
        need_ps = m.args.PS_only or (not m.args.HO_only) or m.args.User_orient_kg_eh
        if need_ps:
            extra += list(memories_h[:max(1, m.p_hop)])
            extra += list(memories_t[:m.p_hop])
        return mark_needed(m.n_entity, m.adj_entity, item_indices, self._depth(), extra)

    def _exchange(self, work, item_indices, memories_h, memories_t):
        if self.is_dense(item_indices.shape[0]):
            self.table.fetch_all(work)
        else:
            self.table.fetch(self.needed(item_indices, memories_h, memories_t), work)

    def forward_device(self, user_indices, item_indices, memories_h, memories_r, memories_t, **kw):
        """Exchange, then score (serialised on the current stream)."""
        self._exchange(self.table.work, item_indices, memories_h, memories_t)
        self.model.entity_emb_matrix = self.table.work
        return self.model.forward_device(user_indices, item_indices, memories_h, memories_r, memories_t, **kw)

    # ---- double-buffered pipeline: exchange for batch i+1 overlaps the scoring of batch i ----
    def enable_pipeline(self):
        self._bufs = [self.table.work, self.table.new_work_table()]
        self._side = torch.cuda.Stream(device=self.model.device, priority=-1)
        self._ready = [None, None]      # event: rows of buffer b are in place (side stream)
        self._free = [None, None]       # event: the scoring that read buffer b is done (main stream)

    def prefetch(self, buf, item_indices, memories_h, memories_t):
        """Start the row exchange for a batch into working table ``buf`` on the side stream."""
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._side):
            if self._free[buf] is None:
                self._side.wait_stream(main)
            else:
                self._side.wait_event(self._free[buf])
            self._exchange(self._bufs[buf], item_indices, memories_h, memories_t)
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._ready[buf] = ev

    def forward_prefetched(self, buf, user_indices, item_indices, memories_h, memories_r, memories_t, **kw):
        """Score a batch whose rows were prefetched into ``buf``."""
        main = torch.cuda.current_stream()
        main.wait_event(self._ready[buf])
        self.model.entity_emb_matrix = self._bufs[buf]
        out = self.model.forward_device(user_indices, item_indices, memories_h, memories_r, memories_t, **kw)
        done = torch.cuda.Event()
        done.record(main)
        self._free[buf] = done
        return out
