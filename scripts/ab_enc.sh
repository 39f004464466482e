#!/bin/bash
# Run ON the GPU box: encoded (packed-tile kernel) vs plain adjacency, config by config -> gpurun_out/ab_enc.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out; out=gpurun_out/ab_enc.txt; : > $out
run() {
  for e in 0 1; do
    line=$(MVIN_L2_ENC=$e python bench.py --no-cpu-baseline --no-hbm-leg --no-sweep --no-probe --steps 10 --warmup 3 "$@" 2>/dev/null | grep '^{' | tail -1)
    python - "$line" "enc=$e $*" >> $out <<'PY'
import json, sys
r = json.loads(sys.argv[1]); t = r["roofline"].get("timed_region", r["roofline"])
print("%-90s step %.4f ms  fused %.4f ms  %.1f Mpairs/s" % (sys.argv[2], r["ms_per_step"], t.get("avg_launch_ms", float("nan")), r["value"] / 1e6))
PY
  done
}
run
run --batch 512
run --batch 4096
run --batch 16384
run --dataset MovieLens-1M --dim 32 --fanout 16 --batch 524288
run --dataset MovieLens-1M --dim 32 --fanout 32 --batch 262144
run --dataset amazon-book_20core --dim 64 --fanout 64 --batch 32768
run --dataset amazon-book_20core --dim 128 --hop 3 --fanout 128 --table-dtype bf16 --batch 64 --steps 3 --warmup 1
run --adj uniform
cat $out
