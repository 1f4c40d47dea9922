cd $GRAFT_REPO_ROOT
bash tools/pmc_bench_traffic.sh r04 > gpurun_out/r17_pmc.log 2>&1
bash tools/make_profiles.sh r04 > gpurun_out/r17_profiles.log 2>&1
