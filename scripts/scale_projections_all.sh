#!/bin/bash
# Run ON the GPU box: the one-GPU projections of the strong-scaling line DESIGN section 6 quotes (bench.py --emulate-world W,
# 100 steps): C3 (scripts/scale_projection.py) and one JSON line per W for C4 at 262 144 / 32 768 pairs and C5 at 512.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
python scripts/scale_projection.py 100
line() {   # <out file> <bench args...>
  out=$1; shift
  : > gpurun_out/$out
  for W in 1 2 4 8; do
    python bench.py --no-hbm-leg --no-sweep --no-cpu-baseline --steps 100 --warmup 5 "$@" $( [ $W -gt 1 ] && echo --emulate-world $W ) 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/$out
  done
  python - gpurun_out/$out <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
t1 = rows[0]["ms_per_step"]
print(sys.argv[1], [(w, round(r["ms_per_step"], 4), round(t1 / (w * r["ms_per_step"]), 3)) for w, r in zip((1, 2, 4, 8), rows)])
PY
}
line scale_projection_c4_B262144.jsonl --dataset amazon-book_20core --dim 64 --fanout 64 --batch 262144
line scale_projection_c4_B32768.jsonl --dataset amazon-book_20core --dim 64 --fanout 64 --batch 32768
line scale_projection_c5_B512.jsonl --dataset amazon-book_20core --dim 128 --hop 3 --fanout 128 --table-dtype bf16 --batch 512
