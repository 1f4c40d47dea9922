#!/bin/bash
# panel-width tables for cfg 3 under the ring kernel (GPC_NB_TABLE: "rem=width,..." -- the first entry with rem >= columns left decides)
run() { echo -n "[$1] "; GPC_NB_TABLE="$1" GPC_BENCH_PHASES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.1f ms  ring %.2f TF x %d  all %.2f TF' % (j['ms_per_step'], r['achieved'], r['launches_per_step'], r['all_trailing_updates']['tflops']))"; }
run ""
run "4096=4096,8192=1024,20480=1024,65536=1536"
run "4096=4096,8192=1024,16384=1024,65536=1536"
run "4096=4096,8192=1024,28672=1024,45056=1536,65536=2048"
run "4096=4096,8192=1024,28672=1024,65536=1792"
run "4096=4096,8192=1024,28672=1280,65536=1536"
run "4096=4096,8192=1024,18432=1024,28672=1536,65536=2048"
run ""
