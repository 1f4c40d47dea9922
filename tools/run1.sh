cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or syrk or potri or chol_inverse or trsm" 2>&1 | tail -15 > gpurun_out/r1_tests.txt
echo "== gemm forms default" > gpurun_out/r1_gemm.txt
python tools/gemm_forms.py >> gpurun_out/r1_gemm.txt 2>&1
echo "== gemm forms KC_PF2=0" >> gpurun_out/r1_gemm.txt
GPC_GEMM_KC_PF2=0 python tools/gemm_forms.py >> gpurun_out/r1_gemm.txt 2>&1
echo "== gemm forms KC_PF2=1" >> gpurun_out/r1_gemm.txt
GPC_GEMM_KC_PF2=1 python tools/gemm_forms.py >> gpurun_out/r1_gemm.txt 2>&1
echo "== gemm forms old generic" >> gpurun_out/r1_gemm.txt
GPC_GEMM_FAST_KC=0 python tools/gemm_forms.py >> gpurun_out/r1_gemm.txt 2>&1
python tools/potri_inplace_ab.py 2048 4096 8192 12288 16384 24576 32768 > gpurun_out/r1_potri.txt 2>&1
GPC_POTRI_LAUUM_NB=2048 python tools/potri_inplace_ab.py 8192 16384 32768 >> gpurun_out/r1_potri.txt 2>&1
python tools/potri_inplace_ab.py 65536 >> gpurun_out/r1_potri.txt 2>&1
GPC_POTRI_LAUUM_NB=2048 python tools/potri_inplace_ab.py 65536 >> gpurun_out/r1_potri.txt 2>&1
