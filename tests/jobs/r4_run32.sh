set -x
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
BIN=$GRAFT_REPO_ROOT/tests/micro/persist_probe
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_wide_$i -- $BIN cold ${PROBE_M:-7168} > $OUT/pmc_wide_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for i in range(1, 5):
    for f in glob.glob(f"{out}/pmc_wide_{i}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
            a = agg[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
with open(f"{out}/r4_pmc_wide_probe.txt", "w") as fo:
    for k in sorted(agg):
        line = f"{k[0]:60s} grid={k[1]:>8s} " + " ".join(f"{c}={v[1]/v[0]:.3g}" for c, v in sorted(agg[k].items()))
        print(line); fo.write(line + "\n")
PY
