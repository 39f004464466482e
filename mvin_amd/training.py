"""Training step of the MVIN path on the GPU (scope row f-2): the reference's
``sess.run([optimizer, loss])`` of model.py:416-417 -- forward, loss of model.py:378-412,
backward, tf.train.AdamOptimizer(lr) update (model.py:414) -- with every arithmetic step in
libmvin_hip.so (mvin_bwd.hip + the forward kernels in their ``_ex`` form that keeps what the
backward needs).  PyTorch only allocates buffers; there is no autograd and no CPU fallback.

The forward used for training is the per-level path (the deepest hop still gathers the K^L
rows without materialising them); each step records a closure that turns the gradient of its
output into gradients of its inputs and parameters, and the closures run in reverse.

Gradient identities used (see DESIGN.md 3.1 for the forward algebra they mirror):
  * attention logits are t[r] = Rel[r].w_r, so dRel / d urh_weights[D:2D] come from the nR-entry
    dT table; the user and self slices of urh_weights only receive their L2 term (exactly what
    autograd gives: the softmax is shift invariant);
  * deepest hop: agg = S'.W_L + (psum/K) c_L with S' = (1/K) sum_k p_k E[y_k], so
    dS' = dZ.W_L^T feeds one gather-form mvin_agg_bwd that scatter-adds (p_k/K) dS' into dE[y_k];
  * key addressing: s_m = h_m . V[b, r_m], V = E[item].R_KGE[r]  =>  dR_KGE[r] = E[item]^T dV[:, r].
"""
import os

import numpy as np
import torch

from . import ops
from .params import aggregator_keys

F32 = torch.float32


class _Grads(object):
    """Gradient accumulators keyed by tensor identity."""

    def __init__(self):
        self.g = {}
        self.keep = []   # keep keyed tensors alive so ids stay unique
        self.lent = set()

    def add(self, t, g, borrowed=False):
        """``borrowed``: ``g`` stays in use (read-only) by the caller, so it must not be accumulated into; it is
        stored as is and replaced by a fresh sum only if another contribution arrives (no copy launch otherwise)."""
        k = id(t)
        if k in self.g:
            if k in self.lent:
                self.g[k] = torch.add(self.g[k], g.view_as(self.g[k]))
                self.lent.discard(k)
            else:
                ops.axpby(1.0, g.view(-1), 1.0, self.g[k].view(-1))
        else:
            self.g[k] = g
            self.keep.append(t)
            if borrowed:
                self.lent.add(k)

    def get(self, t):
        return self.g.get(id(t))


class Trainer(object):
    """Owns the Adam state of an mvin_amd.model.MVIN and runs training steps on it."""

    def __init__(self, model, lr=None, beta1=0.9, beta2=0.999, eps=1e-8, group=None, world=1):
        """``world`` > 1: data-parallel training, one process per GPU.  Every rank steps on its own
        1/world of the batch; the data-dependent gradients (and loss terms) of all ranks are summed
        with ONE all-reduce of the flat gradient buffer (RCCL: torch.distributed backend "nccl") before
        the per-parameter L2 terms and the Adam update, which every rank then applies identically --
        parameters stay bit-identical across ranks."""
        a = model.args
        self.group, self.world = group, int(world)
        if not a.wide_deep:
            raise NotImplementedError("training the legacy aggregate path (wide_deep=False) is not built; the "
                                      "reference cannot run it either (model.py:366-374)")
        if model.entity_emb_matrix.dtype != torch.float32:
            raise NotImplementedError("training needs fp32 tables (table_dtype='bf16' is a scoring-only layout)")
        self.m = model
        self.lr = model.lr if lr is None else lr
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.t = 0
        self.by_entity = os.environ.get("MVIN_TRAIN_BY_ENTITY", "1") != "0"   # de-duplicated deepest-hop backward
        self.batch_wgrads = os.environ.get("MVIN_WGRAD_QUEUE", "1") != "0"     # weight gradients queued -> mvin_linear_wgrad_multi
        self.item_grad_in_kernel_max_batch = 2048    # up to here key_addr_bwd adds dE[item] of V = E[item].R_KGE itself
        self.params = self._named_params()
        self._build_flat_state()
        self.last_grads = None

    def _l2_coefficients(self):
        """Coefficient c of the (c/2) sum(x^2) term of every parameter (slice), model.py:387-412."""
        m, a = self.m, self.m.args
        H, M, P = m.h_hop, m.n_mix_hop, m.p_hop
        L = M * H
        l2w, l2a = float(a.l2_weight), float(a.l2_agg_weight)
        c = {k: 0.0 for k in self.params}
        c["relation_emb_matrix"] = l2w
        lvl = [0.0] * (L + 1)
        if P > 0:
            c["user_mlp_matrix"] = c["user_mlp_bias"] = l2w
            # :407-408 (the first H+1 projections) plus the LAST matrix once more (:405)
            lvl = [l2w * ((1 if e <= H else 0) + (1 if e == L else 0)) for e in range(L + 1)]
        c["transfer_W"] = c["transfer_b"] = lvl
        c["h_emb_item_mlp_matrix"] = c["h_emb_item_mlp_bias"] = l2w
        c["user_emb_matrix"] = l2a
        if not a.PS_only:
            for (i, n) in aggregator_keys(a):
                c[f"agg_{i}_{n}_weights"] = c[f"agg_{i}_{n}_urh_weights"] = l2a
        for n in range(M):
            c[f"enti_transfer_matrix_{n}"] = c[f"enti_transfer_bias_{n}"] = l2a
        return c

    def _build_flat_state(self):
        """Gradients and Adam moments of ALL parameters live in three flat buffers (one zero-fill and
        one optimizer launch per step); a device table maps flat ranges to parameter storage."""
        dev = self.m.device
        coef = self._l2_coefficients()
        segs, off = [], 0
        self._slices = {}
        for k, p in self.params.items():
            self._slices[k] = (off, p.numel())
            cs = coef[k]
            if isinstance(cs, list):           # per-level coefficients of the stacked projections
                per = p.numel() // len(cs)
                for e, ce in enumerate(cs):
                    segs.append((p.data_ptr() + 4 * e * per, off + e * per, per, ce))
            else:
                segs.append((p.data_ptr(), off, p.numel(), cs))
            off += p.numel()
        self._total = off
        self._g = torch.zeros(off, dtype=F32, device=dev)
        self._m = torch.zeros(off, dtype=F32, device=dev)
        self._v = torch.zeros(off, dtype=F32, device=dev)
        view = lambda buf: {k: buf[o:o + n].view_as(self.params[k]) for k, (o, n) in self._slices.items()}
        self._grads, self.adam_m, self.adam_v = view(self._g), view(self._m), view(self._v)
        table = np.zeros(len(segs), dtype=[("x", "<u8"), ("off", "<i8"), ("n", "<i8"), ("l2", "<f4"), ("pad", "<i4")])
        for i, (ptr, o, n, c) in enumerate(segs):
            table[i] = (ptr, o, n, c, 0)
        self._nseg = len(segs)
        self._segs = torch.from_numpy(table.view(np.uint8).copy()).to(dev)

    # ------------------------------------------------------------------ parameters
    def _named_params(self):
        m = self.m
        p = {"user_emb_matrix": m.user_emb_matrix, "entity_emb_matrix": m.entity_emb_matrix,
             "relation_emb_matrix": m.relation_emb_matrix, "relation_emb_KGE_matrix": m.relation_emb_KGE_matrix,
             "user_mlp_matrix": m.user_mlp_matrix, "user_mlp_bias": m.user_mlp_bias,
             "transfer_W": m._transfer_W, "transfer_b": m._transfer_b,
             "h_emb_item_mlp_matrix": m.h_emb_item_mlp_matrix, "h_emb_item_mlp_bias": m.h_emb_item_mlp_bias}
        for n in range(m.n_mix_hop):
            p[f"enti_transfer_matrix_{n}"] = m.enti_transfer_matrix_list[n]
            p[f"enti_transfer_bias_{n}"] = m.enti_transfer_bias_list[n]
        for (i, n), agg in m._agg.items():
            p[f"agg_{i}_{n}_weights"], p[f"agg_{i}_{n}_bias"] = agg.weights, agg.bias
            p[f"agg_{i}_{n}_urh_weights"] = agg.urh_weights
        return p

    def grads_by_reference_name(self):
        """Last step's gradients under the names of mvin_amd/params.py (numpy), for tests."""
        out = {}
        for k, g in self.last_grads.items():
            if k == "transfer_W":
                for e in range(g.shape[0]):
                    out[f"transfer_matrix_{e}"] = g[e].cpu().numpy()
            elif k == "transfer_b":
                for e in range(g.shape[0]):
                    out[f"transfer_bias_{e}"] = g[e].cpu().numpy()
            else:
                out[k] = g.cpu().numpy()
        return out

    # ------------------------------------------------------------------ small helpers
    @staticmethod
    def _T(w):
        return w.t().contiguous()

    def _lookup(self, table, ids):
        return ops.linear([table], None, table.shape[-1], ids=[ids])

    # ------------------------------------------------------------------ one step
    def step(self, user_indices, item_indices, labels, memories_h, memories_r, memories_t, apply=True):
        """One training step on device-resident inputs.  Returns the loss (python float)."""
        return float(self.enqueue(user_indices, item_indices, labels, memories_h, memories_r, memories_t,
                                  apply=apply).item())

    def lr_t(self, t):
        """Bias-corrected step size of tf.train.AdamOptimizer at (1-based) step t."""
        return self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def enqueue(self, user_indices, item_indices, labels, memories_h, memories_r, memories_t, apply=True,
                lr_dev=None):
        """Enqueue one training step on the current stream without any host synchronisation; returns the
        1-element device tensor the loss is accumulated in.  ``lr_dev`` (1-element fp32 device tensor): the
        Adam step size is read from it when the optimizer kernel runs and the step counter is left to the
        caller -- the form GraphedTrainer captures."""
        m, a = self.m, self.m.args
        dev = m.device
        D, K, H, M, P, nR = m.dim, m.n_neighbor, m.h_hop, m.n_mix_hop, m.p_hop, m.n_relation
        L = M * H
        B = item_indices.shape[0]
        E, U, R = m.entity_emb_matrix, m.user_emb_matrix, m.relation_emb_KGE_matrix
        user = user_indices.contiguous()
        item = item_indices.contiguous()
        labels = labels.to(F32).contiguous()
        G = _Grads()
        self._g.zero_()
        dP = self._grads                                                  # parameter gradients (views of _g)
        # every small accumulator of the step out of ONE zero-filled arena (one fill launch instead of six)
        n_agg = len(m._agg)
        arena = torch.zeros(4 + (n_agg + 1) * ((nR + 3) & ~3) + 64 * D, dtype=F32, device=dev)
        loss_acc = arena[0:1]
        arena_off = [4]

        def small_zeros(n):
            o = arena_off[0]
            arena_off[0] = o + ((n + 3) & ~3)
            assert arena_off[0] <= arena.numel()
            return arena[o:o + n]
        tape = []
        # weight gradients are QUEUED and go out together after the tape (mvin_linear_wgrad_multi: problems of one tile
        # shape share a launch): at the reference's batch sizes each is microseconds of work behind ~10 us of launch and
        # ramp, and a step has eleven.  Their inputs are forward tensors and FINAL gradients (a tensor's gradient is
        # complete when its own backward runs; buffers lent to G are never accumulated into), kept alive by the queue.
        wq = []

        def wgrad(*args_, **kw):
            if self.batch_wgrads:
                wq.append(ops.wgrad_problem(*args_, **kw))
            else:
                ops.linear_wgrad(*args_, **kw)

        def zeros(*shape):
            return torch.zeros(shape, dtype=F32, device=dev)

        tcache = {}

        def T3(w):
            """Per-matrix transposes of a [n, D, D] weight stack (or of one [n*D, D] block matrix seen as such):
            ONE launch per stack and step, shared by every backward product that needs W^T."""
            w3 = w.view(-1, D, w.shape[-1]) if w.dim() == 2 else w
            k = (w3.data_ptr(), tuple(w3.shape))
            if k not in tcache:
                tcache[k] = w3.transpose(1, 2).contiguous()
            return tcache[k]

        # ================================================================ forward
        need_ps = a.PS_only or (not a.HO_only) or a.User_orient_kg_eh
        self._ka_bwd_ran = False
        n_o = P + 1 if a.PS_O_ft else P
        ps = None
        if need_ps:
            w_h = m.h_emb_item_mlp_matrix.view(-1) if a.PS_O_ft else None
            V = None
            if P > 0:
                V = torch.empty((B, nR, D), dtype=F32, device=dev)
                ops.linear([E], R, D, ids=[item], rows=B, out=V, ldo=nR * D, nz=nR, w_zstride=D * D, out_zstride=D)
            o_cat = torch.empty((B, n_o * D), dtype=F32, device=dev)
            if ops.key_addressing_supported(m.n_memory, D):
                ops.key_addressing(E, V, w_h, memories_h, memories_r, memories_t, P, o_cat, n_o * D, nR)
            else:
                slot = 0
                if a.PS_O_ft:
                    ops.ripple_attn(E, memories_h[0], None, memories_h[0], None, w_h, 1, o_cat, 0, n_o * D, nR)
                    slot = 1
                for hop in range(P):
                    ops.ripple_attn(E, memories_h[hop], memories_r[hop], memories_t[hop], V, None, 0, o_cat,
                                    (slot + hop) * D, n_o * D, nR)
            ps = ops.linear([o_cat], m.user_mlp_matrix, D, bias=m.user_mlp_bias)

            def bwd_ps():
                d = G.get(ps)
                if d is None:
                    return
                self._ka_bwd_ran = True      # mvin_key_addressing_bwd adds the 2*l2*(h, t) regulariser rows
                wgrad([o_cat], d, dP["user_mlp_matrix"], db=dP["user_mlp_bias"])
                do_cat = torch.empty_like(o_cat)
                # d o_s = d . Wu[sD:(s+1)D]^T for every slot s: one z-batched launch over the transposed blocks
                ops.linear([d], T3(m.user_mlp_matrix), D, out=do_cat, ldo=n_o * D, nz=n_o, w_zstride=D * D, out_zstride=D)
                dV = zeros(B, nR, D) if P > 0 else None
                dw = small_zeros(64 * D).view(64, D) if a.PS_O_ft else None     # replicas: B atomics onto the same D floats serialise
                # ... and, from the rows it reads anyway, their regulariser value l2*(sum h^2 + sum t^2) (model.py:383-385)
                # ... and, where the shape allows, the item's share dE[item] += sum_r dV[:, r] . R[r]^T from the dV block in LDS
                # (measured: -40 us per step at 512 pairs, +20 us at 4 096, where the separate product runs at full width)
                item_in_kernel = (P > 0 and B <= self.item_grad_in_kernel_max_batch
                                  and ops.key_addressing_bwd_adds_item_grad(P, m.n_memory, D, nR))
                ops.key_addressing_bwd(E, V, w_h, memories_h, memories_r, memories_t, P, do_cat, n_o * D, nR,
                                       float(a.l2_weight), dP["entity_emb_matrix"], dV, dw, reg_accum=loss_acc,
                                       relation_kge=R if item_in_kernel else None, items=item if item_in_kernel else None)
                if a.PS_O_ft:
                    dws = torch.empty(D, dtype=F32, device=dev)
                    ops.eltwise(6, D, dw.view(-1), dws, alpha=1.0, D=D, N=dw.shape[0])
                    ops.axpby(1.0, dws, 1.0, dP["h_emb_item_mlp_matrix"].view(-1)[:D])
                if P > 0:
                    # V[b,r,:] = E[item_b] . R[r]  =>  dR[r] += E[item]^T dV[:,r] ; dE[item] += sum_r dV[:,r] R[r]^T
                    wgrad([E], dV, dP["relation_emb_KGE_matrix"], ids=[item], rows=B, nz=nR, ldy=nR * D,
                                     dy_zstride=D, dw_zstride=D * D)
                    if item_in_kernel:
                        return
                    if nR * D <= 4096:   # one product: ditem[b, i] = sum_{r, j} dV[b, r, j] R[r, i, j]
                        ditem = ops.linear([dV.view(B, nR * D)], R.permute(0, 2, 1).reshape(nR * D, D).contiguous(), D)
                    else:
                        ditem = zeros(B, D)
                        for r in range(nR):
                            tmp = ops.linear([dV[:, r, :].contiguous()], self._T(R[r]), D)
                            ops.axpby(1.0, tmp.view(-1), 1.0, ditem.view(-1))
                    ops.scatter_add_rows(dP["entity_emb_matrix"], item, ditem)
            tape.append(bwd_ps)

        u_emb = None
        if a.HO_only or not a.User_orient_kg_eh:
            u_emb = self._lookup(U, user)

            def bwd_u():
                d = G.get(u_emb)
                if d is not None:
                    ops.scatter_add_rows(dP["user_emb_matrix"], user, d)
            tape.append(bwd_u)
        user_o = u_emb if a.HO_only else ps
        q = ps if a.User_orient_kg_eh else u_emb

        dT = {}
        if a.PS_only:
            item_emb = self._lookup(E, item)

            def bwd_item():
                d = G.get(item_emb)
                if d is not None:
                    ops.scatter_add_rows(dP["entity_emb_matrix"], item, d)
            tape.append(bwd_item)
        else:
            ents, rels = m.get_neighbors(item, levels=L - 1)
            Wt, bt = m.transfer_matrix_list, m.transfer_matrix_bias
            ev = []
            c = None
            if a.User_orient:
                # c[e] = q . W_e + b_e for e = 1..L
                c_all = ops.linear([q], m._transfer_W[1:], D, bias=m._transfer_b[1:], nz=L, w_zstride=D * D,
                                   bias_zstride=D, out_zstride=B * D).view(L, B, D)
                c = [c_all[e] for e in range(L)]      # stable tensor objects: gradients are keyed by identity

                def bwd_c():
                    ds = [G.get(c[e - 1]) for e in range(1, L + 1)]
                    for e in range(1, L + 1):
                        if ds[e - 1] is not None:
                            wgrad([q], ds[e - 1], dP["transfer_W"][e], db=dP["transfer_b"][e])
                    if all(d is not None for d in ds) and L <= 4:
                        # dq += sum_e d_e . W_e^T: the concatenated [d_1 | ... | d_L] times the stacked transposes
                        G.add(q, ops.linear(ds, T3(m._transfer_W)[1:].reshape(L * D, D), D))
                    else:
                        for e in range(1, L + 1):
                            if ds[e - 1] is not None:
                                G.add(q, ops.linear([ds[e - 1]], T3(m._transfer_W)[e], D))
                tape.append(bwd_c)
                ev0 = ops.linear([E, q], Wt[0], D, ids=[ents[0].view(-1), None], bias=bt[0], sum_sources=True).view(B, 1, D)

                def bwd_ev0():
                    d = G.get(ev0)
                    if d is None:
                        return
                    d2 = d.view(B, D)
                    wgrad([E, q], d2, dP["transfer_W"][0], ids=[ents[0].view(-1), None], db=dP["transfer_b"][0],
                                     sum_sources=True)
                    dx = ops.linear([d2], T3(m._transfer_W)[0], D)
                    ops.scatter_add_rows(dP["entity_emb_matrix"], ents[0].view(-1), dx)
                    G.add(q, dx, borrowed=True)
                tape.append(bwd_ev0)
                ev.append(ev0)
                for e in range(1, L):
                    ids_e = ents[e].view(-1)
                    t_e = ops.linear([E], Wt[e], D, ids=[ids_e], rowbias=c[e - 1], rows_per_group=K ** e).view(B, -1, D)

                    def bwd_ev(e=e, ids_e=ids_e, t_e=t_e):
                        d = G.get(t_e)
                        if d is None:
                            return
                        d2 = d.view(-1, D)
                        wgrad([E], d2, dP["transfer_W"][e], ids=[ids_e])
                        ops.scatter_add_rows(dP["entity_emb_matrix"], ids_e, ops.linear([d2], T3(m._transfer_W)[e], D))
                        dc = torch.empty((B, D), dtype=F32, device=dev)
                        ops.eltwise(6, B * D, d2, dc, alpha=1.0, D=D, N=K ** e)
                        G.add(c[e - 1], dc)
                    tape.append(bwd_ev)
                    ev.append(t_e)
            else:
                for e in range(L):
                    ids_e = ents[e].view(-1)
                    t_e = self._lookup(E, ids_e).view(B, -1, D)

                    def bwd_raw(ids_e=ids_e, t_e=t_e):
                        d = G.get(t_e)
                        if d is not None:
                            ops.scatter_add_rows(dP["entity_emb_matrix"], ids_e, d.view(-1, D))
                    tape.append(bwd_raw)
                    ev.append(t_e)

            def apply_agg(key, cur, hop, fused):
                agg = m._agg[key]
                N = K ** hop
                T = B * N
                t_tab = agg.relation_scores() if agg.User_orient_rela else None
                if agg.User_orient_rela and key not in dT:
                    dT[key] = small_zeros(nR)
                self_t = cur[hop]
                if fused:
                    Wc = Wt[L] if a.User_orient else None
                    cc = c[L - 1] if a.User_orient else None
                    node_ids = ents[hop].view(-1)
                    out, probs, S, Z = ops.gather_attn_ex(E, m.adj_entity, m.adj_relation, node_ids, t_tab,
                                                          self_t.view(T, D), Wc, cc, agg.weights, agg.bias, B, N, K, D)
                else:
                    child_t = cur[hop + 1]
                    rel_ids = rels[hop].view(-1)
                    out, probs, Z = ops.agg_ex(self_t.view(T, D), child_t.view(T * K, D), rel_ids, t_tab, agg.weights,
                                               agg.bias, B, N, K, D)

                def bwd():
                    d = G.get(out)
                    if d is None:
                        return
                    dm = torch.empty((T, D), dtype=F32, device=dev)
                    ops.eltwise(2, T * D, d.view(-1), dm.view(-1), z=out.view(-1))       # relu'
                    i, n = key
                    wgrad([Z], dm, dP[f"agg_{i}_{n}_weights"], db=dP[f"agg_{i}_{n}_bias"])
                    dZ = ops.linear([dm], T3(agg.weights)[0], D)                         # d(self + neighbors_agg)
                    G.add(self_t, dZ.view_as(self_t), borrowed=True)
                    dTk = dT.get(key)
                    pr = probs.view(T, K) if probs is not None else None
                    if fused:
                        if a.User_orient:
                            psum_over_k = (1.0 / K) if agg.User_orient_rela else 1.0
                            wgrad([S], dZ, dP["transfer_W"][L])               # dW_L += S'^T dZ
                            dS = ops.linear([dZ], T3(m._transfer_W)[L], D)
                            dc = torch.empty((B, D), dtype=F32, device=dev)
                            ops.eltwise(6, B * D, dZ, dc, alpha=psum_over_k, D=D, N=N)
                            G.add(c[L - 1], dc)
                        else:
                            dS = dZ
                        if self.by_entity and (pr is not None or not agg.User_orient_rela):
                            # the deepest hop's backward is linear in dS and depends on a node only through
                            # its entity: sum dS per entity first (T row scatter-adds), then ONE pass per
                            # touched entity instead of one per tree node (a batch repeats entities heavily)
                            Gx = zeros(m.n_entity, D)
                            ops.scatter_add_rows(Gx, node_ids, dS)
                            ops.agg_bwd(Gx, None, m.n_entity, K, D, nR, table=E, adj_entity=m.adj_entity,
                                        adj_relation=m.adj_relation, node_ids=None, rel_score=t_tab,
                                        dtable=dP["entity_emb_matrix"], dT=dTk)
                        else:
                            ops.agg_bwd(dS, pr, T, K, D, nR, table=E, adj_entity=m.adj_entity,
                                        adj_relation=m.adj_relation, node_ids=node_ids,
                                        dtable=dP["entity_emb_matrix"], dT=dTk)
                    else:
                        dchild = ops.agg_bwd(dZ, pr, T, K, D, nR, child=child_t.view(T * K, D), rel_ids=rel_ids, dT=dTk)
                        G.add(child_t, dchild.view_as(child_t))
                tape.append(bwd)
                return out

            item_emb = None
            for n in range(M):
                stages = [ev]
                for i in range(H):
                    nxt = [apply_agg((i, n), ev, hop, fused=(n == 0 and i == 0 and hop == L - 1))
                           for hop in range(L - (H * n + i))]
                    ev = nxt
                    stages.append(ev)
                keep = (M - n - 1) * H + 1
                Wm, bm = m.enti_transfer_matrix_list[n], m.enti_transfer_bias_list[n]
                new = []
                for e in range(keep):
                    srcs = [st[e] for st in stages]
                    res = ops.linear([s_.view(-1, D) for s_ in srcs], Wm, D, bias=bm).view(B, -1, D)

                    def bwd_comb(srcs=srcs, res=res, n=n, Wm=Wm):
                        d = G.get(res)
                        if d is None:
                            return
                        d2 = d.view(-1, D)
                        wgrad([s_.view(-1, D) for s_ in srcs], d2, dP[f"enti_transfer_matrix_{n}"],
                                         db=dP[f"enti_transfer_bias_{n}"])
                        for si, s_ in enumerate(srcs):
                            G.add(s_, ops.linear([d2], T3(Wm)[si], D).view_as(s_))
                    tape.append(bwd_comb)
                    if n == M - 1:
                        item_emb = res
                    else:
                        new.append(res)
                ev = new

        # scores = sum_d user_o * item_emb (model.py:158) ; loss (model.py:379-380)
        _, scores, _ = ops.linear([item_emb.view(B, D)], None, D, score_u=user_o)
        dscore = torch.empty(B, dtype=F32, device=dev)
        # reduce_mean over the GLOBAL batch (model.py:379): with data parallelism the ranks' sums add up
        inv_b = 1.0 / (B * self.world)
        ops.eltwise(1, B, scores, dscore, z=labels, accum=loss_acc, alpha=inv_b, beta=inv_b)
        du = torch.empty((B, D), dtype=F32, device=dev)
        di = torch.empty((B, D), dtype=F32, device=dev)
        ops.eltwise(5, B * D, item_emb.view(B, D), du, z=dscore, alpha=1.0, beta=0.0, D=D)
        ops.eltwise(5, B * D, user_o, di, z=dscore, alpha=1.0, beta=0.0, D=D)
        G.add(user_o, du)
        G.add(item_emb, di.view_as(item_emb))

        # ================================================================ backward
        for fn in reversed(tape):
            fn()
        ops.linear_wgrad_multi(wq)
        del wq[:]
        for (i, n), g in dT.items():   # relation-logit tables -> relation_emb and urh_weights[D:2D]
            ops.rel_score_bwd(m.relation_emb_matrix, m._agg[(i, n)].urh_weights, g, dP["relation_emb_matrix"],
                              dP[f"agg_{i}_{n}_urh_weights"].view(-1))

        # ================================================================ L2 terms (model.py:382-412)
        l2w = float(a.l2_weight)   # the per-parameter terms are added by mvin_l2_adam_multi below
        # gathered-row regulariser (model.py:383-386): sum(h^2) + sum(t^2) + sum(r_emb^2) per hop
        cnt = None
        for hop in range(P):
            for ids in (memories_h[hop], memories_t[hop]):
                flat = ids.reshape(-1)
                # mvin_key_addressing_bwd adds 2*l2*rows AND the value of the term when it ran -- it does not when ps
                # was built but never consumed (HO_only + User_orient_kg_eh without User_orient: q has no consumer)
                if getattr(self, "_ka_bwd_ran", False):
                    continue
                rows = self._lookup(E, flat)
                ops.eltwise(3, rows.numel(), rows.view(-1), accum=loss_acc, alpha=l2w)
                ops.scatter_add_rows(dP["entity_emb_matrix"], flat, rows, alpha=2.0 * l2w)
            # occurrences per relation, without bincount's host sync (and without 262 k atomics on nR addresses);
            # the term is linear in the counts, so the hops share one count vector
            rid = memories_r[hop].reshape(-1)
            if cnt is None:
                cnt = small_zeros(nR)
            if rid.dtype == torch.int32 and rid.is_contiguous() and nR <= 4096:
                ops.count_ids(rid, nR, out=cnt)
            else:
                cnt.scatter_add_(0, rid.long(), torch.ones(rid.shape[0], dtype=F32, device=rid.device))
        if cnt is not None:
            ops.eltwise(5, nR * D * D, R.view(-1), dP["relation_emb_KGE_matrix"].view(-1), z=cnt, alpha=2.0 * l2w,
                        beta=1.0, D=D * D)
            ops.eltwise(7, nR * D * D, R.view(-1), z=cnt, accum=loss_acc, alpha=l2w, D=D * D)   # no host sync

        # per-parameter L2 terms and (apply) the tf.train.AdamOptimizer step (dense; oracle/train_ref.AdamRef):
        # one launch over the flat gradient / moment buffers
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self._g, group=self.group)        # one bucket: every gradient of the model
            dist.all_reduce(loss_acc, group=self.group)
        lr_t = 0.0
        if apply and lr_dev is None:
            self.t += 1
            lr_t = self.lr_t(self.t)
        ops.l2_adam_multi(self._segs, self._nseg, self._total, self._g, self._m, self._v, loss_acc, apply,
                          float(lr_t), self.b1, self.b2, self.eps, lr_dev=lr_dev if apply else None)
        if apply:
            self.m.invalidate()
        self.last_grads = dP
        return loss_acc


class GraphedTrainer(object):
    """hipGraph replay of the training step for launch-bound batch sizes (the reference trains at 512 / 1024,
    src/bash/mvin_*.sh; train.py:46-53 is one ``sess.run`` per batch).

    A step is ~115 short launches (forward in its ``_ex`` form, loss, the backward tape, L2 + Adam) whose cost at
    those sizes is host-side glue, not GPU work.  ``Trainer.enqueue`` is captured ONCE on static input buffers
    (torch.cuda.CUDAGraph = hipGraph on ROCm; the ctypes launches into libmvin_hip.so are recorded on the capture
    stream like any other kernel) and replayed per batch.  What varies between steps stays on the device: the
    batch (static buffers), the parameters and Adam moments (updated in place by the captured optimizer launch)
    and the bias-corrected step size, which ``mvin_l2_adam_multi_dev`` reads from a 1-element device tensor the
    host refills before each replay.  The relation-logit tables are rebuilt inside the graph (the capture starts
    from an invalidated model), so a replay never reads a table of older weights."""

    def __init__(self, trainer, batch_size, ids_dtype=torch.int64, warmup=1):
        if trainer.world > 1:
            raise NotImplementedError("GraphedTrainer captures a single-rank step (the all-reduce stays eager)")
        self.tr = trainer
        m = trainer.m
        dev = m.device
        B, Nm, P = int(batch_size), m.n_memory, max(1, m.p_hop)
        self.users = torch.zeros(B, dtype=ids_dtype, device=dev)
        self.items = torch.zeros(B, dtype=ids_dtype, device=dev)
        self.labels = torch.zeros(B, dtype=F32, device=dev)
        self.mh = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        self.mr = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        self.mt = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        self.lr_dev = torch.zeros(1, dtype=F32, device=dev)
        feed = (self.users, self.items, self.labels, self.mh, self.mr, self.mt)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up outside capture; apply=False leaves every parameter as is
            for _ in range(warmup):
                trainer.enqueue(*feed, apply=False)
        torch.cuda.current_stream().wait_stream(side)
        m.invalidate()                           # the derived tables are to be rebuilt INSIDE the graph
        self.graph = torch.cuda.CUDAGraph()
        from .graph import capture_without_gc
        with capture_without_gc(), torch.cuda.graph(self.graph):
            self.loss = trainer.enqueue(*feed, apply=True, lr_dev=self.lr_dev)
        m.invalidate()                           # nothing ran during capture: no derived table is valid yet
        self._captured = self._storage_key()

    def _storage_key(self):
        """Addresses the captured launches read: a graph outlives neither ``set_adjacency`` nor a parameter tensor
        being REPLACED (in-place updates, which is what the optimizer does, are fine)."""
        m = self.tr.m
        return (m.adj_entity.data_ptr(), m.adj_relation.data_ptr(), m.entity_emb_matrix.data_ptr(),
                m.user_emb_matrix.data_ptr(), m.relation_emb_KGE_matrix.data_ptr())

    def load(self, users, items, labels, mem_h, mem_r, mem_t):
        """Copy a batch (device tensors) into the graph's static input buffers."""
        self.users.copy_(users)
        self.items.copy_(items)
        self.labels.copy_(labels)
        for i in range(min(len(self.mh), len(mem_h))):
            self.mh[i].copy_(mem_h[i])
            self.mr[i].copy_(mem_r[i])
            self.mt[i].copy_(mem_t[i])

    def replay(self):
        """One optimizer step on whatever is in the static buffers; returns the loss as a device tensor
        (no host synchronisation: read it with ``.item()`` when it is wanted)."""
        tr = self.tr
        if self._storage_key() != self._captured:
            raise RuntimeError("the model's adjacency or a parameter tensor was replaced since this step was captured: "
                               "build a new GraphedTrainer")
        tr.t += 1
        self.lr_dev.fill_(float(tr.lr_t(tr.t)))
        self.graph.replay()
        tr.m.invalidate()                        # parameters changed: eager callers rebuild their derived tables
        return self.loss

    def step(self, users, items, labels, mem_h, mem_r, mem_t):
        self.load(users, items, labels, mem_h, mem_r, mem_t)
        return self.replay()
