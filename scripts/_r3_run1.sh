set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest.log
for B in 524288 262144 131072 65536; do
  python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-hbm-leg --no-probe > gpurun_out/r3a/single_$B.json 2>gpurun_out/r3a/single_$B.err
  python bench.py --batch $B --steps 20 --warmup 5 --shard rowshard --force-collectives --no-cpu-baseline --no-sweep --no-hbm-leg --no-probe > gpurun_out/r3a/shard_$B.json 2>gpurun_out/r3a/shard_$B.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r3a/pytest.log
