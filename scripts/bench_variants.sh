#!/bin/bash
# Run ON the GPU box: the bench.py variants quoted in DESIGN.md section 4, one JSON line each
# (with the command that produced it) -> gpurun_out/bench_variants.jsonl
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out; out=gpurun_out/bench_variants.jsonl; : > $out
run() {
  line=$(python bench.py --no-cpu-baseline --no-hbm-leg --no-sweep --steps 10 --warmup 3 "$@" 2>/dev/null | grep '^{' | tail -1)
  [ -n "$line" ] && python - "$line" "$*" >> $out <<'PY'
import json, sys
r = json.loads(sys.argv[1]); r["variant_args"] = sys.argv[2]; print(json.dumps(r))
PY
}
C2="--dataset MovieLens-1M --dim 32 --fanout 16"
C4="--dataset amazon-book_20core --dim 64 --fanout 64 --batch 32768"
C5="--dataset amazon-book_20core --dim 128 --hop 3 --fanout 128 --table-dtype bf16"
run                                             # C3 default (feed: user_triplet_set resident)
run --feed pairs                                # C3, per-pair ripple-set arrays resident
run --batch 262144
run --batch 262144 --adj uniform --items uniform   # worst-case locality
run --batch 16384
run --batch 512
run --batch 512 --graph
run --n-entity 16000000 --batch 32768           # 4 GB table: HBM-bound
run $C2 --batch 524288
run $C4
run $C4 --feed pairs
run $C5 --batch 64 --steps 3 --warmup 1
# the reference's shipped settings (src/bash/mvin_*.sh: dim 16, fan-out 8, depth 2)
S="--dim 16 --fanout 8"
run $S
run $S --batch 512 --feed pairs
run $S --dataset MovieLens-1M
run $S --dataset MovieLens-1M --batch 1024 --feed pairs
run $S --dataset amazon-book_20core
MVIN_L2_D16=0 run $S                             # general fused kernel instead of the wave-per-parent one (A/B)
# projection of the multi-GPU line on one GPU (rank 0's share, user-sorted split, exchange included)
S2="--dim 16 --fanout 8 --mix 2"                     # parser.py:30 default n_mix_hop = 2 with the shipped dims: tree depth 4, 4 681 rows per pair
run $S2 --batch 16384
run $S2 --batch 512 --feed pairs
run --emulate-world 2
run --emulate-world 4
run --emulate-world 8
MVIN_SPLIT_UNR=16 run $C5 --batch 64 --steps 3 --warmup 1     # C5 without the two-batch rotation (A/B)
MVIN_L2_SPLIT=0 run --feed pairs                # symmetric fused kernel (round-1 design) for A/B
MVIN_L2_SPLIT=0 run $C4 --feed pairs
MVIN_L2_SPLIT=0 run $C5 --batch 64 --steps 3 --warmup 1
# entity-table mode (separate mode, own bytes per pair)
run --batch 262144 --hoist cached
run --batch 262144 --hoist step
run --hoist cached --batch 512
run --hoist cached --batch 512 --graph
run --batch 262144 --hoist cached --adj uniform --items uniform
run --hoist cached --n-entity 16000000 --batch 32768
run $C2 --batch 524288 --hoist cached
run $C4 --hoist cached
run $C5 --batch 4096 --steps 3 --warmup 1 --hoist cached
run $C5 --batch 4096 --steps 3 --warmup 1 --hoist step
python - <<'PY'
import json
for l in open("gpurun_out/bench_variants.jsonl"):
    r = json.loads(l)
    print(f'{r["variant_args"]:75s} {r["value"]:14.1f} pairs/s  {r["ms_per_step"]:9.3f} ms  kernel {r["roofline"]["avg_launch_ms"]} ms  {r["roofline"]["achieved"]} GB/s ({r["roofline"]["bound"]})')
PY
