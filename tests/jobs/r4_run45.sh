set -x
mkdir -p gpurun_out
common="--right-context 13 --weights q8_0 --steps 100 --warmup 10 --regions 3 --no-b512 --no-f32-engine --no-host-pcm --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --checkpoint random"
for bd in "64 2" "64 3" "64 4" "96 3" "96 4"; do set -- $bd
  line=$(timeout -k 10 300 python3 bench.py --batch $1 --pipeline-depth $2 $common 2>/dev/null | grep '^{' | tail -n 1)
  echo "batch $1 depth $2: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'])" "$line")" | tee -a gpurun_out/r4_depth_sweep.txt
done
