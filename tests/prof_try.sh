#!/bin/bash
# usage: tests/prof_try.sh <tag> [bench args...]  -- rocprofv3 kernel trace of a short bench run, prints the crash site if any
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --no-buffered "$@" > $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log 2>&1
echo "rc=$? $TAG: $(grep -c SIGSEGV $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log) segv; $(grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log)"
