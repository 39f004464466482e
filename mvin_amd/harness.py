"""Harness counterpart: the callers of the scoring path in the reference
(src/model/MVIN/train.py:112-146, util.py:14-242), restated for the mvin_amd.MVIN model.

Scope row f-3 / f-4 of SURVEY.md section 8: feed assembly, CTR evaluation, top-K evaluation
and the attention case-study dump.  The epoch loop, early stopping and log book-keeping
(train.py:16-109, train_util.py) are not rebuilt.  Behaviours kept on purpose:

  * CTR eval walks FULL batches only and drops the ragged tail (util.py:49); metrics are
    means of per-batch AUC / ACC / F1 (util.py:56);
  * top-K eval scores every (user, item not in the user's train record) pair in batches of
    ``batch_size``, padding the last batch with its last item (util.py:166-177);
  * the NDCG hit list is built with the LAST k of ``k_list`` (the stale loop variable of
    util.py:187-199), i.e. over the top-100 items, then cut at each k inside ndcg_at_k.

``DeviceFeeder`` is the MI355X-first variant of the feed assembly: the ripple sets of all
users live on the GPU once ([n_user, P, 3, Nm] int32) and a batch's memories are gathered there,
so the per-step host work of train.py:117-120 (3*P python lists of B numpy rows) disappears.
"""
import numpy as np


# --------------------------------------------------------------------------- feed assembly
def get_feed_dict(args, model, data, user_triplet_set, start, end):
    """train.py:112-122 / util.py:208-218."""
    feed = {model.user_indices: data[start:end, 0],
            model.item_indices: data[start:end, 1],
            model.labels: data[start:end, 2]}
    for i in range(max(1, args.p_hop)):
        feed[model.memories_h[i]] = [user_triplet_set[u][i][0] for u in data[start:end, 0]]
        feed[model.memories_r[i]] = [user_triplet_set[u][i][1] for u in data[start:end, 0]]
        feed[model.memories_t[i]] = [user_triplet_set[u][i][2] for u in data[start:end, 0]]
    return feed


def get_feed_dict_top_k(args, model, user_list, item, label, user_triplet_set):
    """train.py:124-134 / util.py:220-230."""
    feed = {model.user_indices: user_list, model.item_indices: item, model.labels: label}
    for i in range(max(1, args.p_hop)):
        feed[model.memories_h[i]] = [user_triplet_set[u][i][0] for u in user_list]
        feed[model.memories_r[i]] = [user_triplet_set[u][i][1] for u in user_list]
        feed[model.memories_t[i]] = [user_triplet_set[u][i][2] for u in user_list]
    return feed


def get_user_record(data, is_train=True):
    """train.py:136-146: user -> set of items with label 1."""
    rec = {}
    for u, i, lab in zip(data[:, 0], data[:, 1], data[:, 2]):
        if lab == 1:
            rec.setdefault(u, set()).add(i)
    return rec


class DeviceFeeder(object):
    """Ripple sets resident on the GPU; batches assembled by device-side row selection."""

    def __init__(self, model, user_triplet_set):
        import torch
        self.model = model
        if torch.is_tensor(user_triplet_set):   # already [n_user, P, 3, n_memory] (data_prep.get_user_triplet_set)
            self.uts = user_triplet_set.to(model.device).to(torch.int32).contiguous()
            self.P = self.uts.shape[1]
            return
        arr = np.asarray(user_triplet_set) if not isinstance(user_triplet_set, dict) else None
        if arr is None:  # dict user -> [P,3,Nm] (the reference's defaultdict)
            n_user = model.n_user
            some = next(iter(user_triplet_set.values()))
            arr = np.zeros((n_user,) + np.asarray(some).shape, dtype=np.int32)
            for u, v in user_triplet_set.items():
                arr[u] = v
        self.model = model
        self.uts = torch.from_numpy(np.ascontiguousarray(arr.astype(np.int32))).to(model.device)
        self.P = self.uts.shape[1]

    def memories(self, users_dev):
        sel = self.uts[users_dev.long()]
        return ([sel[:, i, 0].contiguous() for i in range(self.P)],
                [sel[:, i, 1].contiguous() for i in range(self.P)],
                [sel[:, i, 2].contiguous() for i in range(self.P)])

    def scores(self, users, items):
        """sigmoid scores of (users[i], items[i]) as a device tensor; inputs numpy or tensors."""
        import torch
        dev = self.model.device
        u = torch.as_tensor(np.asarray(users) if not torch.is_tensor(users) else users).to(dev).long()
        it = torch.as_tensor(np.asarray(items) if not torch.is_tensor(items) else items).to(dev).long()
        return self.model.forward_users(u, it, self.uts).scores_normalized     # feeds assembled inside the kernels


    def scores_user(self, user, items):
        """sigmoid scores of ONE user against ``items``: the shared-user form of the path (the
        user's ripple sets are read once per call, not once per pair)."""
        import torch
        dev = self.model.device
        it = torch.as_tensor(np.asarray(items) if not torch.is_tensor(items) else items).to(dev).long()
        sel = self.uts[int(user)]                                   # [P, 3, Nm]
        u = torch.full((1,), int(user), dtype=torch.int64, device=dev)
        return self.model.forward_device(u, it, [sel[i, 0].contiguous() for i in range(self.P)],
                                         [sel[i, 1].contiguous() for i in range(self.P)],
                                         [sel[i, 2].contiguous() for i in range(self.P)]).scores_normalized


# --------------------------------------------------------------------------- training loop
def train_epoch(args, model, train_data, user_triplet_set, sess=None, rng=None):
    """One epoch of train.py:56-64 through ``model.train(sess, feed_dict)``: shuffle, then full
    minibatches only (the ragged tail is skipped, train.py:60-62).  Returns the list of losses."""
    (rng or np.random).shuffle(train_data)
    losses, start = [], 0
    while start + args.batch_size <= train_data.shape[0]:
        _, loss = model.train(sess, get_feed_dict(args, model, train_data, user_triplet_set, start,
                                                  start + args.batch_size))
        losses.append(loss)
        start += args.batch_size
    return losses


def train_epoch_device(feeder, train_data, batch_size, rng=None, graph=False):
    """Same epoch with device-side feeds (ripple sets gathered on the GPU).  ``graph=True``: every step is one
    hipGraph replay (training.GraphedTrainer, captured once per batch size) and the losses are read back once,
    at the end of the epoch, instead of once per step."""
    import torch
    from .training import GraphedTrainer, Trainer
    model = feeder.model
    if model.trainer is None:
        model.trainer = Trainer(model)
    gt = None
    if graph and model.trainer.world == 1:     # a data-parallel step keeps its all-reduce eager (GraphedTrainer refuses)
        gt = getattr(model, "_graphed_trainer", None)
        if (gt is None or gt.tr is not model.trainer or gt.users.shape[0] != batch_size
                or gt._storage_key() != gt._captured):      # set_adjacency / a rebound parameter: capture again
            try:
                gt = model._graphed_trainer = GraphedTrainer(model.trainer, batch_size)
            except RuntimeError as e:                       # capture failed: the same kernels, launched eagerly
                import warnings
                warnings.warn(f"hipGraph capture of the training step failed ({e}); running the epoch eagerly")
                gt = model._graphed_trainer = None
    (rng or np.random).shuffle(train_data)
    dev = model.device
    data = torch.from_numpy(np.ascontiguousarray(train_data)).to(dev)
    losses, start = [], 0
    while start + batch_size <= data.shape[0]:
        blk = data[start:start + batch_size]
        users, items, labels = blk[:, 0].contiguous(), blk[:, 1].contiguous(), blk[:, 2].to(torch.float32)
        mh, mr, mt = feeder.memories(users)
        if gt is not None:
            losses.append(gt.step(users, items, labels, mh, mr, mt).clone())
        else:
            losses.append(model.trainer.step(users, items, labels, mh, mr, mt))
        start += batch_size
    if gt is not None and losses:
        losses = torch.cat(losses).cpu().tolist()
    return losses


# --------------------------------------------------------------------------- ranking metrics
def precision_at_k(ranked, answers, k):
    """metrics.py:33-48: |top-k intersect answers| / k."""
    return len(set(ranked[:k]) & set(answers)) / k


def recall_at_k(ranked, answers, k):
    """metrics.py:96-111: |top-k intersect answers| / |answers|."""
    return len(set(ranked[:k]) & set(answers)) / len(answers)


def dcg_at_k(r, k):
    """metrics.py:3-18 with method=1: sum_i r_i / log2(i + 2)."""
    r = np.asarray(r, dtype=np.float64)[:k]
    return float(np.sum(r / np.log2(np.arange(2, r.size + 2)))) if r.size else 0.0


def ndcg_at_k(r, k):
    """metrics.py:21-31."""
    best = dcg_at_k(sorted(r, reverse=True), k)
    return dcg_at_k(r, k) / best if best else 0.0


# --------------------------------------------------------------------------- evaluations
def ctr_eval(args, model, data, user_triplet_set, batch_size, sess=None):
    """util.py:44-56 through the reference-shaped ``model.eval(sess, feed_dict)``."""
    aucs, accs, f1s = [], [], []
    start = 0
    while start + batch_size <= data.shape[0]:
        auc, acc, f1 = model.eval(sess, get_feed_dict(args, model, data, user_triplet_set, start, start + batch_size))
        aucs.append(auc)
        accs.append(acc)
        f1s.append(f1)
        start += batch_size
    return aucs, accs, f1s, float(np.mean(aucs)), float(np.mean(accs)), float(np.mean(f1s))


def eval_streams(model, batch_size):
    """How many HIP streams the independent batches of an evaluation go round-robin to (measured at C3): TWO where a pass is the
    single launch (<= MVIN.small_max_batch pairs: 24.3 us per 512-pair batch against 36.0 on one stream and 28.8 on three), THREE
    for the multi-launch schedule (4 096 pairs: 71 vs 76 us on two; 524 288: 1.058 vs 1.082 ms; four: 1.097)."""
    return 2 if batch_size <= int(getattr(model, "small_max_batch", 0) or 0) else 3


def ctr_eval_device(feeder, data, batch_size, streams=None, window=16):
    """Same numbers as ctr_eval, feeds assembled on the device.  The batches of an evaluation are independent: they are
    enqueued round-robin on ``streams`` HIP streams (None: ``eval_streams`` -- at the reference's batch sizes a scoring pass is one
    launch that walks a dependent chain of loads, two passes in flight overlap their waits: 24 vs 36 us per 512-pair batch at C3),
    ``window`` batches ahead of the host-side metrics; ``streams=1`` scores and reads back one batch at a time."""
    import torch
    from sklearn.metrics import f1_score, roc_auc_score
    aucs, accs, f1s = [], [], []
    starts = list(range(0, data.shape[0] - batch_size + 1, batch_size))
    dev = feeder.model.device
    from .graph import scoring_streams
    if streams is None:
        streams = eval_streams(feeder.model, batch_size)
    lanes = scoring_streams(dev, streams) if streams > 1 else None
    if lanes:
        # everything the forward builds lazily and CACHES (relation-logit tables dropped by invalidate() after a training epoch,
        # the adjacency encoding, the one-time id check and the static per-user records of user_triplet_set) is built here, on
        # the current stream, before the lanes start: a table built by the first batch on lane 0 would be read by the second
        # batch on lane 1 with nothing ordering the two
        feeder.model.prepare(getattr(feeder, "uts", None))
        for ln in lanes:
            ln.wait_stream(torch.cuda.current_stream(dev))
    pending = []

    def finish(entry):
        blk, s, ev = entry
        if ev is not None:
            ev.synchronize()
        s = s.cpu().numpy()
        labels = blk[:, 2].astype(np.float32)
        aucs.append(roc_auc_score(y_true=labels, y_score=s))
        pred = (s >= 0.5).astype(np.float32)
        f1s.append(f1_score(y_true=labels, y_pred=pred))
        accs.append(float(np.mean(pred == labels)))

    for i, start in enumerate(starts):
        blk = data[start:start + batch_size]
        if lanes:
            with torch.cuda.stream(lanes[i % streams]):
                s = feeder.scores(blk[:, 0], blk[:, 1])
                ev = torch.cuda.Event()
                ev.record()
        else:
            s, ev = feeder.scores(blk[:, 0], blk[:, 1]), None
        pending.append((blk, s, ev))
        if len(pending) > (window if lanes else 0):
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))
    return aucs, accs, f1s, float(np.mean(aucs)), float(np.mean(accs)), float(np.mean(f1s))


def topk_settings(train_data, eval_data, test_data, n_item, user_num=250, k_list=(1, 2, 5, 10, 25, 50, 100)):
    """util.py:14-41 without the pickle round trip: the ``user_num`` users with the most
    positive train interactions among those present in all three splits."""
    train_record = get_user_record(train_data, True)
    test_record = get_user_record(test_data, False)
    eval_record = get_user_record(eval_data, False)
    users = list(set(train_record) & set(test_record) & set(eval_record))
    users = sorted(users, key=lambda u: len(train_record[u]), reverse=True)[:user_num]
    return users, train_record, eval_record, test_record, set(range(n_item)), list(k_list)


def _rank_metrics(item_sorted, truth, k_list, precision_list, recall_list, ndcg_list):
    for k in k_list:
        precision_list[k].append(precision_at_k(item_sorted, truth, k))
        recall_list[k].append(recall_at_k(item_sorted, truth, k))
    k_stale = k_list[-1]  # util.py:193 uses the loop variable left over from the loop above
    r_hit = [1 if i in truth else 0 for i in item_sorted[:k_stale]]
    for k in k_list:
        ndcg_list[k].append(ndcg_at_k(r_hit, k))


def topk_eval(args, user_triplet_set, model, user_list, train_record, eval_record, test_record, item_set,
              k_list, batch_size, mode="test", sess=None):
    """util.py:137-205 through ``model.get_scores(sess, feed_dict)``."""
    precision_list = {k: [] for k in k_list}
    recall_list = {k: [] for k in k_list}
    ndcg_list = {k: [] for k in k_list}
    ref = eval_record if mode == "eval" else test_record
    for user in user_list:
        if user not in ref:
            continue
        test_items = list(item_set - train_record[user])
        score_of = {}
        start = 0
        while start + batch_size <= len(test_items):
            items, scores = model.get_scores(sess, get_feed_dict_top_k(
                args, model, [user] * batch_size, test_items[start:start + batch_size], [1] * batch_size,
                user_triplet_set))
            score_of.update(zip(items, scores))
            start += batch_size
        if start < len(test_items):  # pad the ragged tail with its last item (util.py:166-177)
            pad = test_items[start:] + [test_items[-1]] * (batch_size - len(test_items) + start)
            items, scores = model.get_scores(sess, get_feed_dict_top_k(
                args, model, [user] * batch_size, pad, [1] * batch_size, user_triplet_set))
            score_of.update(zip(items, scores))
        item_sorted = [i for i, _ in sorted(score_of.items(), key=lambda x: x[1], reverse=True)]
        _rank_metrics(item_sorted, ref[user], k_list, precision_list, recall_list, ndcg_list)
    return ([float(np.mean(precision_list[k])) for k in k_list], [float(np.mean(recall_list[k])) for k in k_list],
            [float(np.mean(ndcg_list[k])) for k in k_list], None, None)


def topk_eval_device(feeder, user_list, train_record, eval_record, test_record, item_set, k_list,
                     batch_size, mode="test"):
    """top-K evaluation with device-side feeds.  Every (user, candidate item) pair is scored
    exactly once (no padded duplicates: the scoring path takes any batch length); ranking
    ties are broken like the reference's ``sorted`` (stable, by insertion order)."""
    precision_list = {k: [] for k in k_list}
    recall_list = {k: [] for k in k_list}
    ndcg_list = {k: [] for k in k_list}
    ref = eval_record if mode == "eval" else test_record
    for user in user_list:
        if user not in ref:
            continue
        test_items = np.fromiter(item_set - train_record[user], dtype=np.int64)
        scores = np.empty(len(test_items), dtype=np.float32)
        for start in range(0, len(test_items), batch_size):
            blk = test_items[start:start + batch_size]
            scores[start:start + len(blk)] = feeder.scores_user(user, blk).cpu().numpy()
        order = np.argsort(-scores, kind="stable")
        _rank_metrics(test_items[order].tolist(), ref[user], k_list, precision_list, recall_list, ndcg_list)
    return ([float(np.mean(precision_list[k])) for k in k_list], [float(np.mean(recall_list[k])) for k in k_list],
            [float(np.mean(ndcg_list[k])) for k in k_list], None, None)


# --------------------------------------------------------------------------- the train.py loop
class EarlyStop(object):
    """train_util.py:20-61 (Early_stop_info): keep the best evaluation score, save the stage-wise
    tables whenever it improves (``save_final_model``), stop after ``early_stop`` epochs without
    improvement once ``tolerance`` epochs have passed."""

    def __init__(self, tolerance=2, early_stop=3, save_final_model=True):
        self.best, self.best_saved = -float("inf"), -float("inf")
        self.counter, self.tolerance, self.early_stop = 0, tolerance, early_stop
        self.save_final_model = save_final_model

    def update(self, epoch, score, model):
        if score > self.best_saved:
            self.best_saved = score
            if self.save_final_model and model.path is not None and getattr(model.path, "emb", None):
                model.save_pretrain_emb_fuc(None, None)
        if epoch + 1 > self.tolerance:
            if score > self.best:
                self.best, self.counter = score, 0
            else:
                self.counter += 1
            if self.counter >= self.early_stop:
                return True
        return False


def train(args, data, show_topk=False, model=None, device="cuda", rng=None, log=None, topk_batch=65536, hoist=True,
          topk_early_stop=False, graph="auto"):
    """train.py:16-109 on the GPU path.  ``data`` = the 16-tuple of mvin_amd.data_io.load_data / the
    reference's ``load_data`` (read by position exactly as train.py:17-21 does; a 10-tuple prefix
    (..., user_triplet_set) is accepted for CTR runs).  Per epoch: shuffle, full minibatches only
    (:56-64), then CTR evaluation on train/eval/test with early stopping on the eval AUC (:87-103) or,
    with ``show_topk``, top-K evaluation over the candidate set ``item_set_most_pop`` = data[12]
    (:69-77 pass it where topk_eval's signature says item_set) scored on eval recall@k_list[2].
    In top-K mode the reference NEVER stops early: :86 compares update_score's return value
    ('EarlyStopping' or None) with True.  That is reproduced (the best-epoch save still happens);
    ``topk_early_stop=True`` applies the evident intent instead (listed in INTEGRATION.md section 4).
    ``hoist``: evaluate through the entity-table mode (weights are frozen while an epoch is evaluated; the
    tables are dropped by every optimizer step and rebuilt by the first evaluation batch) -- 2-4x faster
    evaluation at the same tolerance; training steps always take the faithful kernels.
    ``graph``: replay every optimisation step as one hipGraph (training.GraphedTrainer; same kernels, losses read back
    once per epoch); "auto" = at batch sizes up to 2 048, where a step is launch- and latency-bound (the reference's
    scripts train at 512 / 1 024).
    Returns (model, history): one dict per epoch."""
    from .model import MVIN
    n_user, n_item, n_entity, n_relation = data[0], data[1], data[2], data[3]
    train_data, eval_data, test_data = (np.asarray(d) for d in data[4:7])
    adj_entity, adj_relation, uts = data[7], data[8], data[9]
    if model is None:
        model = MVIN(args, n_user, n_entity, n_relation, adj_entity, adj_relation, device=device, hoist=bool(hoist))
        if getattr(args, "load_pretrain_emb", False):
            model.restore_pretrain_emb()                                       # train.py:53-54
    feeder = DeviceFeeder(model, uts)
    stop = EarlyStop(getattr(args, "tolerance", 2), getattr(args, "early_stop", 3),
                     getattr(args, "save_final_model", True))
    if show_topk:
        user_list, train_rec, eval_rec, test_rec, item_set, k_list = topk_settings(train_data, eval_data, test_data,
                                                                                   n_item)
        if len(data) > 12 and data[12] is not None:
            item_set = set(int(i) for i in data[12])                            # item_set_most_pop, train.py:70,75
    history = []
    train_data = train_data.copy()
    for epoch in range(getattr(args, "n_epochs", 20)):
        losses = train_epoch_device(feeder, train_data, args.batch_size, rng=rng,
                                    graph=(args.batch_size <= 2048) if graph == "auto" else bool(graph))
        rec = {"epoch": epoch, "loss": float(np.mean(losses)) if losses else float("nan")}
        if show_topk:
            for mode in ("eval", "test"):
                p, r, n, _, _ = topk_eval_device(feeder, user_list, train_rec, eval_rec, test_rec, item_set, k_list,
                                                 topk_batch, mode=mode)
                rec[mode] = {"precision": p, "recall": r, "ndcg": n}
            score = rec["eval"]["recall"][2]
        else:
            for name, d in (("train", train_data), ("eval", eval_data), ("test", test_data)):
                _, _, _, auc, acc, f1 = ctr_eval_device(feeder, d, args.batch_size)
                rec[name] = {"auc": auc, "acc": acc, "f1": f1}
            score = rec["eval"]["auc"]                                          # Eval_score_info.eval_st_score
        history.append(rec)
        if log:
            log(rec)
        if stop.update(epoch, score, model) and (topk_early_stop or not show_topk):
            break
    return model, history


# --------------------------------------------------------------------------- case study (f-4)
def _names(ids, table):
    return [table[str(i)] if str(i) in table else str(i) for i in ids]


def ctr_eval_case_study(args, model, data, user_triplet_set, user_history_dict, entity_index_2_name,
                        rela_index_2_name, user_list, item_set_most_pop, batch_size, path, sess=None):
    """util.py:59-127: dump sampled neighbors, relations and attention weights per selected
    (user, item) pair in the reference's text layout.  Written to ``path`` (the reference derives
    the file name from args.path.case_st / log_name / epoch / SW_stage)."""
    nb = args.neighbor_sample_size
    star20, star50 = "*" * 20, "*" * 50
    with open(path, "w") as f:
        f.write(f"{star50}\n case_study \n")
        start = 0
        while start + batch_size <= data.shape[0]:
            users, labels, items, ents, rels, imp0, imp1 = model.eval_case_study(
                sess, get_feed_dict(args, model, data, user_triplet_set, start, start + batch_size))
            for b in range(batch_size):
                if users[b] not in user_list or items[b] not in item_set_most_pop:
                    continue
                f.write(f"{star50}\n")
                f.write(f"user_indices = {users[b]}, item_indices = {items[b]}, labels = {labels[b]}\n")
                f.write(f"{star20} first_layer  {star20}\n")
                f.write(f"et_index 0 = {','.join(str(x) for x in ents[0][b, :].tolist())}\n")
                f.write(f"rela_index 0 = {','.join(str(x) for x in rels[0][b, :].tolist())}\n")
                f.write(f"et_index 1 = {','.join(str(x) for x in ents[1][b, :].tolist())}\n")
                f.write(f"{star20} second_layer  {star20}\n")
                for k in range(nb):
                    f.write(f"entities 0 = {k}\n")
                    f.write(f"et_index 0 = {ents[1][b, k].tolist()}\n")
                    f.write(f"rela_index 0 = {','.join(str(x) for x in rels[1][b, nb * k: nb * (k + 1)].tolist())}\n")
                    f.write(f"et_index 1 = {','.join(str(x) for x in ents[2][b, nb * k: nb * (k + 1)].tolist())}\n")
                f.write(f"{star20} entity_relation_name  {star20}\n")
                hist = _names(user_history_dict.get(users[b], []), entity_index_2_name)
                f.write(f"item_name = {_names([items[b]], entity_index_2_name)[0]}\n")
                f.write(f"user_interact_items = {','.join(hist)}\n")
                ename = [_names(e[b, :], entity_index_2_name) for e in ents]
                rname = [_names(r[b, :], rela_index_2_name) for r in rels]
                f.write(f"{star20} first_layer  {star20}\n")
                f.write(f"et_index 0 = {','.join(ename[0])}\n")
                pairs = ["rela = %s, enti = %s, att = %s" % p for p in zip(rname[0], ename[1], imp0[b, :][0])]
                f.write("er rela pair 0 = " + "\n".join(pairs) + "\n")
                f.write(f"{star20} second_layer  {star20}\n")
                for k in range(nb):
                    f.write(f"entities 0 = {k}\n")
                    f.write(f"et_index 0 = {ename[1][k]}\n")
                    pairs = ["rela = %s, enti = %s, att = %s" % p for p in
                             zip(rname[1][nb * k: nb * (k + 1)], ename[2][nb * k: nb * (k + 1)], imp1[b, k])]
                    f.write("er rela pair 1 = " + "\n".join(pairs) + "\n")
            start += batch_size
