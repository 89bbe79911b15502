#!/bin/bash
# round 3, cycle 27: headline fill: 64-row chopping (old) vs whole-run spans with three ways of warming L2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), 'alone', round(d['valu_roofline']['launch_ms_alone'],3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2; do for lib in old warm1 warm0 warm2; do
  WARPSTR_HIP_LIB=$R/build/exp/lib$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c27_b.json 2> $O/r03c27_b.err || { tail $O/r03c27_b.err; exit 1; }
  line $O/r03c27_b.json "headline [$lib]"
done; done
