# full GPU suite + smoke + determinism soak (GPU box)
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r46_fullsuite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r46_fullsuite.txt 2>&1
( python tools/repeat_check.py 8192 16384 32768 2>&1 | grep -v amdgpu.ids; python tools/flow_soak.py 100 2>&1 | tail -2 ) >> gpurun_out/r46_fullsuite.txt 2>&1
