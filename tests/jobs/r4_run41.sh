set -x
mkdir -p gpurun_out
common="--right-context 13 --weights q8_0 --steps 60 --warmup 10 --regions 3 --no-b512 --no-f32-engine --no-host-pcm --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --checkpoint random"
for b in 512 128; do for d in 2 3 4; do
  line=$(timeout -k 10 300 python3 bench.py --batch $b --pipeline-depth $d $common 2>/dev/null | grep '^{' | tail -n 1)
  echo "batch $b depth $d: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'])" "$line")" | tee -a gpurun_out/r4_depth_sweep.txt
done; done
