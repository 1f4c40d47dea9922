cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests/ -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r6_fullsuite.txt 2>&1
