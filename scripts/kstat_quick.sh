cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lm && rocprofv3 --kernel-trace --stats -d /tmp/prof_lm -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-hbm-leg --no-sweep --no-probe --steps 10 --warmup 3 > /dev/null 2>&1
python - <<PY
import csv,glob
fs=glob.glob("/tmp/prof_lm/**/*kernel_stats.csv", recursive=True)
print(fs[:2])
for r in list(csv.DictReader(open(fs[0])))[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
