cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r04
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r59_fullsuite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r59_fullsuite.txt 2>&1
for i in 1 2 3 4 5; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "evaluations\|wall" | tail -2; done > gpurun_out/profiles_r04/gplvm_cfg5.txt 2>&1
python bench.py --workload cfg2 --steps 20 --warmup 3 > gpurun_out/profiles_r04/bench_cfg2.json 2> /dev/null
for i in 1 2 3 4 5 6 7 8; do python -m pytest tests/test_dtc.py -q -m gpu 2>&1 | tail -1; done >> gpurun_out/r59_fullsuite.txt
