#!/bin/bash
# usage: tests/micro/trace_overlap.sh <tag> ["ENV=1 ENV2=2"] [bench args...]   (on the GPU box)
# rocprofv3 --kernel-trace of PIPELINED steps of 64 streams x R = 13 (Q8_0 tensors), then the overlap structure of the trace:
# how many GEMM launches run at a time, how long they take when they share the chip, how the lanes' starts line up.
TAG=$1; ENVS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--batch 64 --right-context 13 --weights q8_0 --checkpoint random --no-b512 --no-f32-engine --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --regions 1 --steps ${PROF_STEPS:-16} --warmup 3 $*"
cd /tmp && export TMPDIR=/tmp
for kv in $ENVS; do export $kv; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace_$TAG.log 2>&1
echo "trace pass rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/trace_$TAG.log | head -1)"
python3 $GRAFT_REPO_ROOT/tests/micro/trace_overlap.py $OUT/trace_$TAG $OUT/overlap_$TAG.json
rm -rf $OUT/trace_$TAG
