cd $GRAFT_REPO_ROOT
for per in 16 32 48 96 192 384 1024; do echo -n "per=$per  "; GPC_GRAM_PER=$per python tools/gram_bench.py 65536 32 2>/dev/null; done > gpurun_out/r11_per.txt 2>&1
for per in 48 192 1024; do echo -n "per=$per  "; GPC_GRAM_PER=$per python tools/gram_bench.py 65536 8 2>/dev/null; done >> gpurun_out/r11_per.txt 2>&1
