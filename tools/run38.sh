cd $GRAFT_REPO_ROOT
for W in 512 1024 1536 2048 4096; do
  for D in 8 32; do
    echo -n "PAIR_WGS=$W " >> gpurun_out/r38_gram.txt
    GPC_GRAM_PAIR_WGS=$W python tools/gram_bench.py 65536 $D 2>/dev/null >> gpurun_out/r38_gram.txt
  done
done
python tools/gplvm_time.py 2>/dev/null | tail -3 >> gpurun_out/r38_gram.txt
