cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_grid_gpu.py -x -q -k "gram or kern_fixt or compute or tiles or cfg3 or synthetic" 2>&1 | tail -6 > gpurun_out/r9_tests.txt
for D in 32 16 8 4; do python tools/gram_bench.py 65536 $D 2>/dev/null; done > gpurun_out/r9_gram.txt 2>&1
python tools/gram_bench.py 8192 8 >> gpurun_out/r9_gram.txt 2>/dev/null
