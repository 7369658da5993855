set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r4_counters_avail.txt 2>&1
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*" $GRAFT_REPO_ROOT/gpurun_out/r4_counters_avail.txt | sort -u
