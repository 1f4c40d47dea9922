cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_grid_gpu.py -x -q -k "gram or kern_fixt or compute or tiles or cfg3 or synthetic" 2>&1 | tail -4 > gpurun_out/r12_tests.txt
for wgs in 512 1024 2048 4096; do for D in 32 8; do echo -n "pair_wgs=$wgs "; GPC_GRAM_PAIR_WGS=$wgs python tools/gram_bench.py 65536 $D 2>/dev/null; done; done > gpurun_out/r12_gram.txt 2>&1
for N in 8192 16384 32768 131072; do python tools/gram_bench.py $N 16 2>/dev/null; GPC_GRAM_PAIRS=0 python tools/gram_bench.py $N 16 2>/dev/null | sed 's/$/ [pairs off]/'; done >> gpurun_out/r12_gram.txt 2>&1
