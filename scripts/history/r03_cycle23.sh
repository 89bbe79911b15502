#!/bin/bash
# round 3: what the driver runs at round end -- the GPU suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/r03c23_gpu_tests.log 2>&1 || { tail -60 $O/r03c23_gpu_tests.log; exit 1; }
tail -1 $O/r03c23_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
start=$(date +%s.%N)
timeout -k 10 600 python bench.py > $O/r03c23_bench.json 2> $O/r03c23_bench.err || { tail -20 $O/r03c23_bench.err; exit 1; }
end=$(date +%s.%N); echo "bench.py wall: $(python3 -c "print(round($end-$start,1))") s"
python3 -c "
import json; d=json.load(open('$O/r03c23_bench.json'))
print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['launch_ms'], d['verified']['mismatches'], {k: (round(v.get('value', v.get('reads_per_s_hbm_int16',0))), v.get('verified',{}).get('mismatches')) for k,v in d['secondary'].items()})"
