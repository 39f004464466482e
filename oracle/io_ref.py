"""Loop restatements (TEST INFRASTRUCTURE) of the reference's on-disk loading rules, written the way
the reference writes them -- Python dicts and per-row loops -- to check the vectorised
mvin_amd/data_io.py against.  Follows src/model/MVIN/data_loader_user_set.py: load_rating (:33-110),
load_pre_data (:241-254), load_kg (:276-289)."""
import csv

import numpy as np


def read_pd_csv_ref(path):
    """:242-244: read_csv, drop the first column, select ['user','item','like']."""
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    header, body = rows[0], rows[1:]
    idx = {name: i for i, name in enumerate(header)}
    return np.array([[int(r[idx["user"]]), int(r[idx["item"]]), int(r[idx["like"]])] for r in body], dtype=np.int64)


def most_popular_items_ref(rating_np, top_k):
    """:49-57."""
    item_count = {}
    for i in range(rating_np.shape[0]):
        item = int(rating_np[i, 1])
        if item not in item_count:
            item_count[item] = 0
        item_count[item] += 1
    ranked = sorted(item_count.items(), key=lambda x: x[1], reverse=True)[:top_k]
    return set(k for k, _ in ranked)


def user_history_ref(train_data):
    """:78-88."""
    hist = {}
    for i in range(train_data.shape[0]):
        user, item, rating = (int(x) for x in train_data[i])
        if rating == 1:
            if user not in hist:
                hist[user] = []
            hist[user].append(item)
    return hist


def filter_known_users_ref(data, hist):
    """:90-96."""
    keep = [i for i in range(data.shape[0]) if int(data[i][0]) in hist]
    return data[keep]


def kg_counts_ref(kg_np):
    """:286-287."""
    return len(set(kg_np[:, 0]) | set(kg_np[:, 2])), len(set(kg_np[:, 1]))
