#!/bin/bash
# Step time of the batch call at the per-GPU shares of a strong-scaled configs[3] (100k reads over 1/2/4/8 GPUs), on ONE GPU.
# Usage: scripts/share_sweep.sh TAG    -> gpurun_out/TAG_share_sweep.log (+ a one-stream kernel breakdown of the 12.5k share)
TAG=$1; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_share_sweep.log
: > $L
for n in 100000 50000 25000 12500; do
  steps=$((2000000 / n)); [ $steps -gt 100 ] && steps=100
  out=$(timeout -k 10 200 python $R/bench.py --no-cpu-baseline --reads $n --steps $steps --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s  called_ok %d' % (d['ms_per_step'], d['value'], d['config']['called_ok']))") || exit 1
  echo "reads $n steps $steps : $out" | tee -a $L
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_share_prof -o p -- python3 $R/bench.py --no-cpu-baseline --reads 12500 --steps 20 --warmup 2 > $R/gpurun_out/${TAG}_share_prof.log 2>&1 || exit 1
python3 $R/scripts/kstats.py $R/gpurun_out/${TAG}_share_prof/p_kernel_trace.csv 23 | tee -a $L
python3 $R/scripts/trace_overlap.py $R/gpurun_out/${TAG}_share_prof/p_kernel_trace.csv | tee -a $L
