cd $GRAFT_REPO_ROOT
( python tools/repeat_check.py 8192 16384 32768 49152 2>&1 | grep -v amdgpu.ids; python tools/flow_soak.py 100 2>&1 | tail -3 ) > gpurun_out/r27_soak.txt 2>&1
python tools/gemm_forms.py 2>/dev/null > gpurun_out/r27_gemm_forms.txt
