#!/bin/bash
# One GPU cycle: parity tests, bench line, single-stream per-kernel trace.  Usage: scripts/gpu_cycle.sh TAG
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG}_gpu_tests.log 2>&1 || { tail -30 $R/gpurun_out/${TAG}_gpu_tests.log; exit 1; }
tail -1 $R/gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 300 python bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail $R/gpurun_out/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$R/gpurun_out/${TAG}_bench.json'))
print('reads/s', d['value'], 'ms/step', d['ms_per_step'], 'fill alone ms', d['valu_roofline']['launch_ms_alone'])
PY
cd /tmp && export TMPDIR=/tmp
WSX_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_prof.log 2>&1
python3 $R/scripts/kstats.py $R/gpurun_out/${TAG}_prof/p_kernel_trace.csv 5
