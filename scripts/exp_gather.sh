# bench, plain and through the collective path (one-rank group), with the HIP runtime's hardware-queue limit raised or not
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for q in 4 8 16; do
a=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python $R/bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['called_ok'])")
b=$(GPU_MAX_HW_QUEUES=$q WARPSTR_BENCH_SELF_GATHER=1 timeout -k 10 200 python $R/bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
echo "hwq=$q plain $a   self-gather $b"
done; done
