cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "pair_walk or gradx or grad_cross" 2>&1 | tail -15 > gpurun_out/r4_tests.txt
python -m pytest tests/test_dtc.py tests/test_gpu_parity.py -x -q -k "dtc or gplvm or potri_in_place" 2>&1 | tail -5 >> gpurun_out/r4_tests.txt
for d in 4 8 16 32; do python tools/grad_bench.py 32768 $d 2>&1 | grep gradx; GPC_PAIR_WALK=0 python tools/grad_bench.py 32768 $d 2>&1 | grep gradx | sed 's/$/   [scalar]/'; done > gpurun_out/r4_gradx.txt
python tools/dtc_bench.py > gpurun_out/r4_dtc.txt 2>&1
GPC_PAIR_WALK=0 python tools/dtc_bench.py >> gpurun_out/r4_dtc.txt 2>&1
python tools/potri_inplace_ab.py 32768 65536 > gpurun_out/r4_potri.txt 2>&1
