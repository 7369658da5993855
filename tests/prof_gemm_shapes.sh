#!/bin/bash
# usage: tests/prof_gemm_shapes.sh   (on the GPU box, through gpurun)
# Per-shape durations of the large-M GEMM kernels at 64 streams x R = 13 (rocprofv3 --kernel-trace, grouped by kernel and grid)
# beside the vendor library's kernels for the same shapes (tests/micro/blaslt_ref.py): -> gpurun_out/gemm_shapes.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --regions 1 --sync-steps --steps 12 --warmup 2 --batch 64 --right-context 13 --weights q8_0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gs_engine -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/gs_blas -- python3 $GRAFT_REPO_ROOT/tests/micro/blaslt_ref.py > /dev/null 2>&1
python3 - <<PY | tee $OUT/gemm_shapes.txt
import csv, glob, collections
def load(d):
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm" not in n and "Cijk" not in n: continue
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
        acc[(n[:60], g, int(r.get("LDS_Block_Size", 0) or 0))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc
for title, d in (("engine, 64 streams x R = 13 (M = 896), synchronous steps", "/tmp/gs_engine"), ("vendor library through torch.matmul, M = 896 (W1, W2, QKV, pw1, Wo/pw2 in that order of size)", "/tmp/gs_blas")):
    print("==", title)
    for (n, g, lds), v in sorted(load(d).items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 20: continue
        v.sort()
        print(f"{n:60s} grid {str(g):16s} lds {lds:6d} calls {len(v):5d} avg {sum(v)/len(v):6.2f} us  p50 {v[len(v)//2]:6.2f}")
PY
