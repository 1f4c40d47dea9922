# regenerates the committed rNN profiles on the GPU box: PMC traffic of the bench command first (bench.py replays it), the
# default and cfg 2 lines, kernel traces and counters (make_profiles.sh, pmc_kgrad.sh), gradient / gemm / DTC / GP-LVM timings
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles_$TAG
bash tools/pmc_bench_traffic.sh $TAG > gpurun_out/regen_pmc_bench.log 2>&1
cp gpurun_out/profiles_$TAG/pmc_bench_traffic.json profiles/${TAG}_pmc_bench_traffic.json
python bench.py > gpurun_out/profiles_$TAG/bench_default.json 2> gpurun_out/regen_bench_default.err
GPC_BENCH_GRID=1 python bench.py --no-cpu-baseline > gpurun_out/profiles_$TAG/bench_cfg3_grid_1x1.json 2> gpurun_out/regen_bench_grid.err
python tools/grid_costs.py gpurun_out/profiles_$TAG/grid_costs.json > gpurun_out/regen_grid_costs.log 2>&1
python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/profiles_$TAG/bench_cfg4_1gpu.json 2> gpurun_out/regen_bench_cfg4.err
python tools/posterior_bench.py 65536 1024 > gpurun_out/profiles_$TAG/posterior_cfg3.txt 2>&1
python bench.py --workload cfg2 --steps 20 --warmup 3 > gpurun_out/profiles_$TAG/bench_cfg2.json 2> gpurun_out/regen_bench_cfg2.err
bash tools/make_profiles.sh $TAG > gpurun_out/regen_make_profiles.log 2>&1
bash tools/pmc_kgrad.sh $TAG > gpurun_out/regen_pmc_kgrad.log 2>&1
for D in 4 8 16 32; do python tools/grad_bench.py 65536 $D 2>/dev/null | grep "kern_grad"; done > gpurun_out/profiles_$TAG/grad_timings.txt 2>&1
python tools/gemm_forms.py 2>/dev/null > gpurun_out/profiles_$TAG/gemm_forms.txt
python tools/dtc_bench.py 2>/dev/null | tail -5 > gpurun_out/profiles_$TAG/dtc.txt
for i in 1 2 3; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "evaluations\|seconds\|took" | tail -2; done > gpurun_out/profiles_$TAG/gplvm_cfg5.txt 2>&1
