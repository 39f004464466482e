"""Aggregator layer -- the reference's call surface (src/model/MVIN/aggregators.py:17-152)
on top of libmvin_hip.so.

    SumAggregator_urh_matrix(save_model_name, batch_size, dim, dropout=0., act=relu,
                             name=None, User_orient_rela=True)
    agg(self_vectors, neighbor_vectors, neighbor_relations, user_embeddings, masks)
        -> (relu((self + mix(neigh)) . weights + bias) [B,N,D],  probs [B,N,K] or None)

Tensors are torch CUDA(ROCm) fp32 tensors.  ``neighbor_relations`` may be what the
reference passes -- relation VECTORS [B,N,K,D] (aggregators.py:130) -- or, the form the
MVIN model of this package uses, int32 relation IDS [B,N,K] together with
``relation_emb`` given to the constructor / ``bind_relation_table`` (the logits are then
looked up in an nR-entry table instead of gathered per child).
"""
from abc import abstractmethod

import torch

from . import ops

LAYER_IDS = {}


def get_layer_id(layer_name=""):
    """aggregators.py:8-14 (unused by MVIN, kept for surface parity)."""
    if layer_name not in LAYER_IDS:
        LAYER_IDS[layer_name] = 0
        return 0
    LAYER_IDS[layer_name] += 1
    return LAYER_IDS[layer_name]


def relu(x):
    """Stand-in for tf.nn.relu in constructor signatures; the activation is always the
    fused ReLU of the kernel epilogue (aggregators.py:96 overrides the ctor argument)."""
    return torch.relu(x)


class Aggregator(object):
    """aggregators.py:17-35."""

    def __init__(self, save_model_name, batch_size, dim, dropout, act, name):
        layer = self.__class__.__name__.lower()
        self.name = layer + "_" + save_model_name + "_" + str(name)
        self.dropout = dropout
        self.act = act
        self.batch_size = batch_size
        self.dim = dim

    def __call__(self, self_vectors, neighbor_vectors, neighbor_relations, user_embeddings, masks):
        return self._call(self_vectors, neighbor_vectors, neighbor_relations, user_embeddings, masks)

    @abstractmethod
    def _call(self, self_vectors, neighbor_vectors, neighbor_relations, user_embeddings, masks):
        pass

    def _mix_neighbor_vectors(self, neighbor_vectors, neighbor_relations, user_embeddings):
        """aggregators.py:37-56 (KGCN's mixer: p = softmax_k(mean_d(user * relation)), mean_k(p * neighbor)).  Defined by the
        reference and never called by MVIN; built for the completeness of the class surface (mvin_mix_neighbor_vectors_fwd).
        neighbor_vectors / neighbor_relations [B,N,K,D] device tensors, user_embeddings [B,D] -> [B,N,D]."""
        B = neighbor_vectors.shape[0]
        nv = neighbor_vectors.contiguous()
        return ops.mix_neighbor_vectors(nv, neighbor_relations.contiguous().view_as(nv), user_embeddings.contiguous().view(B, self.dim))

    _mix_neighbor_vectors_urv = _mix_neighbor_vectors


class SumAggregator_urh_matrix(Aggregator):
    """aggregators.py:79-152."""

    def __init__(self, save_model_name, batch_size, dim, dropout=0., act=relu, name=None,
                 User_orient_rela=True, weights=None, bias=None, urh_weights=None, urh_bias=None,
                 relation_emb=None, device=None, seed=1):
        super().__init__(save_model_name, batch_size, dim, dropout, act, name)
        if dropout != 0.:
            raise NotImplementedError("dropout != 0 is never used by the reference (aggregators.py:80,109)")
        device = device or (weights.device if isinstance(weights, torch.Tensor) else "cuda")
        if weights is None:
            # aggregators.py:83-93: xavier(seed=1) weights / urh_weights, zero biases
            from .params import xavier_uniform
            import numpy as np
            rng = np.random.default_rng(seed)
            weights = xavier_uniform(rng, (dim, dim))
            urh_weights = xavier_uniform(rng, (3 * dim, 1))
            bias = np.zeros(dim, np.float32)
            urh_bias = np.zeros(1, np.float32)

        def dev(x):
            return torch.as_tensor(x, dtype=torch.float32).to(device).contiguous()

        self.weights = dev(weights)
        self.bias = dev(bias)
        self.urh_weights = dev(urh_weights)
        self.urh_bias = dev(urh_bias)  # created, never added (aggregators.py:92-93 vs :133)
        self.User_orient_rela = User_orient_rela
        self.act = relu
        self._relation_emb = relation_emb
        self._t_cache = None

    # ------------------------------------------------------------------ helpers
    def bind_relation_table(self, relation_emb):
        self._relation_emb = relation_emb
        self._t_cache = None

    def invalidate(self):
        """Call after changing urh_weights / the relation table in place."""
        self._t_cache = None

    def relation_scores(self):
        """t[r] = relation_emb[r] . urh_weights[D:2D]: the only k-dependent part of the logits
        of aggregators.py:130-133 (user and self terms are constant over k and cancel)."""
        if self._t_cache is None:
            if self._relation_emb is None:
                raise ValueError("relation ids given but no relation table bound")
            self._t_cache = ops.rel_score(self._relation_emb, self.urh_weights)
        return self._t_cache

    # ------------------------------------------------------------------ reference surface
    def _call(self, self_vectors, neighbor_vectors, neighbor_relations, user_embeddings, masks):
        """aggregators.py:98-116.  Returns (output [B,N,D], probs_normalized [B,N,K] | None)."""
        D = self.dim
        B, N = self_vectors.shape[0], self_vectors.shape[1]
        K = neighbor_vectors.shape[2]
        self_vectors = self_vectors.contiguous()
        neigh = neighbor_vectors.contiguous().view(B * N * K, D)
        rel_ids = logits = None
        if self.User_orient_rela:
            if neighbor_relations.dtype in (torch.int32, torch.int64):
                rel_ids = neighbor_relations.to(torch.int32).contiguous().view(-1)
                logits = self.relation_scores()
            else:
                # reference form: relation vectors [B,N,K,D] -> one logit per child
                rv = neighbor_relations.contiguous().view(B * N * K, D)
                w_r = self.urh_weights[D:2 * D].contiguous()
                logits = ops.linear([rv], w_r, 1).view(-1)
        out, probs = ops.agg(self_vectors.view(B * N, D), neigh, rel_ids, logits, self.weights,
                             self.bias, B, N, K, D, want_probs=self.User_orient_rela)
        return out, probs

    def _mix_neighbor_vectors_urh(self, self_vectors, user_embeddings, neighbor_vectors, neighbor_relations):
        """aggregators.py:118-146 on its own -> (neighbors_aggregated [B,N,D], probs_normalized [B,N,K]).  ``_call`` does not
        come through here (it runs the fused mvin_agg_fwd / mvin_gather_attn_fwd); this is the standalone form of the class
        surface.  The logit of child k is [user ; relation_k ; self] . urh_weights: the user and self terms do not depend on k
        and cancel in the softmax, so only relation_k . urh_weights[D:2D] is computed.  ``neighbor_relations``: relation
        vectors [B,N,K,D] (the reference's form) or int relation ids [B,N,K]."""
        D = self.dim
        nv = neighbor_vectors.contiguous()
        B, N, K = nv.shape[0], nv.shape[1], nv.shape[2]
        if neighbor_relations.dtype in (torch.int32, torch.int64):
            rv = ops.linear([self._relation_emb], None, D, ids=[neighbor_relations.to(torch.int32).contiguous().view(-1)])
        else:
            rv = neighbor_relations.contiguous().view(B * N * K, D)
        logits = ops.linear([rv], self.urh_weights[D:2 * D].contiguous(), 1).view(B, N, K)
        return ops.mix_neighbor_vectors(nv, want_probs=True, logits=logits)

    def _mix_neighbor_vectors_no_ur(self, self_vectors, user_embeddings, neighbor_vectors, neighbor_relations):
        """aggregators.py:148-152 on its own: the plain mean over the K neighbours."""
        return ops.mix_neighbor_vectors(neighbor_vectors.contiguous())
