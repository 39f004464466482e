"""hipGraph replay of the scoring path for launch-bound batch sizes.

At the reference's own batch sizes (512 / 1024, src/bash/mvin_*.sh) one scoring pass is ~10 short
kernels: launch latency, not the GPU, sets the step time.  ``GraphedScorer`` captures
``MVIN.forward_device`` once on static input buffers (torch.cuda.CUDAGraph = hipGraph on ROCm; the
ctypes launches into libmvin_hip.so are recorded like any other kernel on the capture stream) and
replays it per batch: one graph launch instead of ~10 kernel launches plus Python glue.
"""
import contextlib
import gc

import torch

_STREAMS = {}


@contextlib.contextmanager
def capture_without_gc():
    """Around a hipGraph capture: collect first, then keep the cyclic collector OFF until the capture ends.  A capture runs
    plain Python (hundreds of short-lived objects per step); a collection that starts inside it may free device tensors other
    streams still hold (``record_stream``: the allocator records events on those streams) or destroy stream / event objects
    -- none of which is allowed while a stream captures: the process aborts (seen once in a full test run, in the garbage
    collector under ``torch.cuda.current_stream()``)."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def scoring_streams(device, n=3):
    """The process's first ``n`` streams for independent scoring passes, from ONE pool per device that only grows.  HIP maps streams
    onto a handful of hardware queues round-robin and two streams that share a queue serialise: a pair of streams created late in a
    process that already holds several landed on one queue (the two-stream rate at 512 pairs was 36 us instead of 24).  Callers that
    pipeline independent batches (mvin_amd.harness.ctr_eval_device, bench.py) take their streams from here; asking for two and later
    for three returns the same first two."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


class GraphedScorer(object):
    def __init__(self, model, batch_size, warmup=2):
        self.model = model
        dev = model.device
        B, Nm, P = batch_size, model.n_memory, max(1, model.p_hop)
        self.users = torch.zeros(B, dtype=torch.int64, device=dev)
        self.items = torch.zeros(B, dtype=torch.int64, device=dev)
        self.mh = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        self.mr = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        self.mt = [torch.zeros((B, Nm), dtype=torch.int32, device=dev) for _ in range(P)]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up outside capture (fills the relation-logit caches)
            for _ in range(warmup):
                model.forward_device(self.users, self.items, self.mh, self.mr, self.mt)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with capture_without_gc(), torch.cuda.graph(self.graph):
            self.out = model.forward_device(self.users, self.items, self.mh, self.mr, self.mt)
        # the graph reads derived tables (relation logits, hoisted entity tables) that an optimizer step,
        # set_adjacency or restore_pretrain_emb replaces: such a graph must be captured again
        self._generation = getattr(model, "_generation", 0)

    def load(self, users, items, mem_h, mem_r, mem_t):
        """Copy a batch (device tensors) into the graph's static input buffers."""
        self.users.copy_(users)
        self.items.copy_(items)
        for i in range(len(self.mh)):
            self.mh[i].copy_(mem_h[i])
            self.mr[i].copy_(mem_r[i])
            self.mt[i].copy_(mem_t[i])

    def replay(self):
        """Score whatever is in the static buffers; returns the (static) output namespace."""
        if getattr(self.model, "_generation", 0) != self._generation:
            raise RuntimeError("the model's parameters / adjacency changed since this graph was captured: "
                               "build a new GraphedScorer")
        self.graph.replay()
        return self.out

    def __call__(self, users, items, mem_h, mem_r, mem_t):
        self.load(users, items, mem_h, mem_r, mem_t)
        return self.replay()
