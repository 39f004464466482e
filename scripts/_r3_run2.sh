set -x
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r3b/pytest.log
for W in 2 4 8; do
  python bench.py --emulate-world $W --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-hbm-leg --no-probe > gpurun_out/r3b/emu_$W.json 2>gpurun_out/r3b/emu_$W.err
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-hbm-leg --no-probe > gpurun_out/r3b/emu_1.json 2>gpurun_out/r3b/emu_1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'], r['config'].get('key_addressing_variant','')[:12])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r3b/pytest.log
