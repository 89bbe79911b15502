#!/bin/bash
# The 60 000-read from_fast5 run N times in fresh processes (second leg of each process: the first one of a process starts late),
# with its timeline.  Usage (GPU box): scripts/exp_ff_runs.sh OUT_DIR [runs] [readers] ; env is passed on (A/B switches)
out=${1:-gpurun_out/ff_runs}; n=${2:-3}; readers=${3:-14}; mkdir -p "$out"
for rep in $(seq 1 $n); do
  WARPSTR_BENCH_READER_SWEEP=$readers,$readers WARPSTR_BENCH_TIMELINE=1 WARPSTR_BENCH_FAST5_ONLY=reader_sweep timeout -k 10 400 python scripts/exp_from_fast5.py 1500 > "$out/run_$rep.json" 2> "$out/run_$rep.err" || exit 1
  python - "$out/run_$rep.json" "$readers" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))['reader_sweep'][sys.argv[2]]
ev = {}
for l in r['timeline']:
    name = l.split(None, 1)[1].split(' [cpu')[0]
    ev.setdefault(name.split(' (its ')[0], float(l.split()[0]))
    if name.startswith('batch') and 'answered' in name and int(name.split()[1]) in (2, 3, 4, 5, 20):
        print('   ', l.split(' [cpu')[0])
print(round(r['reads_per_s']), 'wall', round(r['wall_s'], 3), 'returns', round(r['call_returns_after_s'], 3), {k: ev.get(k) for k in
      ('the streamed run begins', 'part set up', 'batch 0 handed to the readers (64 reads)', 'set-up done', 'batch 2 answered', 'batch 5 answered', 'batch 10 answered', 'last batch collected', 'handle closed', 'outputs written')}, flush=True)
PY
done
