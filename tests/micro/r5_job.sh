cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROF_MODE=--sync-steps PROF_STEPS=20 bash tests/prof_r5.sh b64_R13 --batch 64 --right-context 13 --weights q8_0
