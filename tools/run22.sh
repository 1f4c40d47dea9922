cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dtc.py tests/test_host_layer.py -x -q 2>&1 | tail -3 > gpurun_out/r22_tests.txt
python tools/dtc_bench.py > gpurun_out/r22_dtc.txt 2>&1
python tools/dtc_bench.py 65536 1024 8 dtcvar >> gpurun_out/r22_dtc.txt 2>&1
python tools/dtc_bench.py 65536 1024 8 fitc >> gpurun_out/r22_dtc.txt 2>&1
