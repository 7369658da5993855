cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_buf -- python3 $GRAFT_REPO_ROOT/bench.py --chunks-per-step 256 --sync-steps --checkpoint random --no-b512 --no-f32-engine --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --regions 1 --steps 10 --warmup 2 > $OUT/prof_buf.log 2>&1
echo rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/prof_buf.log | head -1)
python3 - <<PY
import csv, glob
fs = glob.glob("$OUT/prof_buf/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(fs[0])))
for r in rows[:22]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
rm -rf $OUT/prof_buf
