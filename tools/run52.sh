cd $GRAFT_REPO_ROOT
python tools/potri_inplace_ab.py 24576 32768 40960 49152 65536 2>/dev/null > gpurun_out/r52.txt
