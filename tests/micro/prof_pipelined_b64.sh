cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b64p -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --right-context 13 --weights q8_0 --checkpoint random --no-b512 --no-f32-engine --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --regions 1 --steps 16 --warmup 3 > $OUT/prof_b64p.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob
fs = glob.glob("$OUT/prof_b64p/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(fs[0])))
with open("$OUT/b64p_kernel_stats.md", "w") as f:
    f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write(f"| \`{r['Name'][:90]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
print(open("$OUT/b64p_kernel_stats.md").read())
PY
rm -rf $OUT/prof_b64p
