#!/bin/bash
# usage: tests/prof_diar_pmc.sh      (on the GPU box, through gpurun)
# HBM-side traffic of the diarization side-car's kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section:
# the two counters do not fit one pass; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read: read bytes = 2 x FETCH_SIZE KiB, write bytes = WRITE_SIZE KiB; the
# counters tally L2 -> fabric requests, Infinity-Cache hits included).  -> gpurun_out/r6_pmc_traffic_diar.json: per kernel, and summed over the launches of ONE bf16 embedding call.
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_diar_$C -- python3 $GRAFT_REPO_ROOT/tests/micro/diar_bench.py > $OUT/pmc_diar_$C.log 2>&1
  echo "$C pass rc=$? segv=$(grep -c SIGSEGV $OUT/pmc_diar_$C.log)"
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
agg = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/pmc_diar_{C}/*/*counter_collection.csv")
    if not fs:
        print(C, "no counter file"); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != C: continue
        a = acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nasr::", "")]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        agg[k][C + "_per_launch"] = round(v / n, 1); agg[k]["launches"] = n
for k, d in agg.items():
    if "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
        d["hbm_bytes_per_launch_corrected"] = int(2 * d["FETCH_SIZE_per_launch"] * 1024 + d["WRITE_SIZE_per_launch"] * 1024)
# launches of ONE bf16 embedding call of 96 sub-segments (profiles/r6_diar_kernel_stats.md): kernel -> launches per call
per_call = {"k_spk_gemm<0, 1>": 6, "k_spk_gemm<1, 1>": 4, "k_spk_gemm<2, 1>": 3, "k_spk_gemm<3, 0>": 1, "k_spk_gemm<1, 0>": 1, "k_diar_logmel": 1, "k_spk_front": 1,
            "k_spk_tile<0>": 1, "k_spk_tile<1>": 1, "k_spk_fc": 12, "k_spk_fc_reduce": 7, "k_gemm_tiled3": 1, "k_spk_att_post": 1}
call_bytes = sum(agg[k]["hbm_bytes_per_launch_corrected"] * n for k, n in per_call.items() if "hbm_bytes_per_launch_corrected" in agg.get(k, {}))
missing = [k for k in per_call if "hbm_bytes_per_launch_corrected" not in agg.get(k, {})]
json.dump({"command": "tests/prof_diar_pmc.sh  (rocprofv3 --pmc <C> --kernel-trace, one counter per pass; tests/micro/diar_bench.py)",
           "units": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read), write bytes = WRITE_SIZE x 1024; L2 -> fabric requests, Infinity-Cache hits included",
           "embed_call_96_segments": {"hbm_bytes_corrected": call_bytes, "launches_per_call": per_call, "kernels_without_counters": missing},
           "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1].get("launches", 0)))},
          open(f"{out}/r6_pmc_traffic_diar.json", "w"), indent=1)
print("embedding call:", call_bytes / 1e6, "MB", "missing:", missing)
for k, n in per_call.items():
    d = agg.get(k, {})
    print(f"{k:22s} x{n:2d}  per launch {d.get('hbm_bytes_per_launch_corrected', 0) / 1e6:8.2f} MB (read {2 * d.get('FETCH_SIZE_per_launch', 0) * 1024 / 1e6:7.2f}, write {d.get('WRITE_SIZE_per_launch', 0) * 1024 / 1e6:7.2f})")
PY
rm -rf $OUT/pmc_diar_FETCH_SIZE $OUT/pmc_diar_WRITE_SIZE
