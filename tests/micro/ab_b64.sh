#!/bin/bash
# A/B of an engine option (or an environment variable) on 64 streams x R = 13 from Q8_0 tensors (BASELINE configs[2]), pipelined and synchronous.
# usage: tests/micro/ab_b64.sh "opt:gemm_cores=0" "opt:persistent_gemm=0" ["VAR=value" ...]   (first run: the build's defaults)
common="--batch ${AB_BATCH:-64} --right-context ${AB_R:-13} --weights q8_0 --steps 100 --warmup 10 --regions 3 --no-b512 --no-f32-engine --no-host-pcm --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --checkpoint ${AB_CHECKPOINT:-random}"
run() {
    local tag="$1"; shift
    local opts="" envs="NASR_AB=0"
    for kv in "$@"; do case "$kv" in opt:*) opts="$opts --engine-option ${kv#opt:}";; *) envs="$envs $kv";; esac; done
    for mode in "" "--sync-steps"; do
        [ -n "$mode" ] && [ "${AB_SYNC:-1}" = 0 ] && continue          # AB_SYNC=0: pipelined steps only
        local line
        line=$(env $envs timeout -k 10 300 python3 bench.py $common $opts $mode 2>/dev/null | grep '^{' | tail -n 1)
        echo "$tag ${mode:-pipelined}: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], 'ms', d['value'], 'RTFx')" "$line")"
    done
}
run base ${AB_BASE:-NASR_AB=0}
for kv in "$@"; do run "$kv" $kv; done
