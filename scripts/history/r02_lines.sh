#!/bin/bash
# Bench lines of round 2 (strict: counters must come from profiles/fill_pmc.json).  Usage: scripts/r02_lines.sh TAG
R=$GRAFT_REPO_ROOT; TAG=$1; O=$R/gpurun_out
python bench.py --from-raw > $O/${TAG}_bench_headline.json 2> $O/${TAG}_bench_headline.err || { tail $O/${TAG}_bench_headline.err; exit 1; }
for w in cfg1 cfg5; do python bench.py --workload $w > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail $O/${TAG}_bench_$w.err; exit 1; }; done
for f in headline cfg1 cfg5; do python3 -c "
import json; d=json.load(open('$O/${TAG}_bench_$f.json'))
print('$f', round(d['value']), round(d['ms_per_step'],3), 'roofline', round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'valu', round(d['valu_roofline']['frac'],3), round(d['valu_roofline']['frac_at_observed_clock'],3), 'alone', round(d['valu_roofline']['launch_ms_alone'],3), 'verified', d['verified']['reads'], d['verified']['mismatches'], 'cpu', round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d.get('from_raw'))"; done
scripts/share_sweep.sh $TAG | head -4
WARPSTR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --scaling strong --no-cpu-baseline > $O/${TAG}_bench_strong2_gloo.json 2> $O/${TAG}_bench_strong2_gloo.err || { tail $O/${TAG}_bench_strong2_gloo.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/${TAG}_bench_strong2_gloo.json')); print('strong 2 ranks on one GPU (gloo)', round(d['value']), round(d['ms_per_step'],3), d['config']['workload'][:90], d['config']['called_ok'])"
