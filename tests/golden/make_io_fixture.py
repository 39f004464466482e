"""Generate tests/golden/io_ml1m_eval_pd_head.{csv,npy} (run from the repo root, in the build
container where /root/reference exists):

    python tests/golden/make_io_fixture.py

The .csv is DATA: the header and the first 64 rows of the reference's own
data/MovieLens-1M/eval_pd.csv; the .npy is what the reference's load_pre_data recipe
(data_loader_user_set.py:246-248: pandas read_csv, drop column 0, select ['user','item','like'])
returns for it, computed here with pandas itself."""
import os

import numpy as np
import pandas as pd

SRC = "/root/reference/data/MovieLens-1M/eval_pd.csv"
OUT = os.path.dirname(os.path.abspath(__file__))

with open(SRC) as f:
    lines = [next(f) for _ in range(65)]
dst = os.path.join(OUT, "io_ml1m_eval_pd_head.csv")
with open(dst, "w") as f:
    f.writelines(lines)
d = pd.read_csv(dst, index_col=None)
d = d.drop(d.columns[0], axis=1)
np.save(os.path.join(OUT, "io_ml1m_eval_pd_head.npy"), d[["user", "item", "like"]].values.astype(np.int64))
print("wrote", dst)
