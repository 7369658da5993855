set -x
mkdir -p gpurun_out
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke_final.txt 2>&1; echo rc $?; tail -3 gpurun_out/r4_smoke_final.txt
