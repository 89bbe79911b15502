mkdir -p gpurun_out/r06c
for cfg in "3 2" "4 2" "6 2" "3 4" "4 4" "6 4" "4 8"; do set -- $cfg; export WARPSTR_ARENA_REGIONS=$1 WARPSTR_CHUNKS_PER_READER=$2; WARPSTR_BENCH_FAST5_ONLY=reader_processes_4x_the_copies timeout -k 10 300 python scripts/exp_from_fast5.py 1500 > gpurun_out/r06c/sw_$1_$2.json 2>> gpurun_out/r06c/sw.err; echo "$cfg rc=$?"; done
