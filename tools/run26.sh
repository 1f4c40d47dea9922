cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r26_fullsuite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r26_smoke.txt 2>&1
