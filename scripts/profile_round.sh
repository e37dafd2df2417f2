#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench, PMC passes (separate runs), and the bench line itself.
# usage (on the GPU box, from the repo root): scripts/profile_round.sh r01g
set -e
cd "$(dirname "$0")/.."
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$tag
python3 bench.py --steps 5 --warmup 1 > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/bench.err
tail -1 gpurun_out/prof_$tag/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/prof_$tag/trace_bench.json 2> gpurun_out/prof_$tag/trace.err
find gpurun_out/prof_$tag/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$tag/kernel_stats.csv \;
head -12 gpurun_out/prof_$tag/kernel_stats.csv | cut -c1-200
python3 scripts/pmc.py gpurun_out/prof_$tag/pmc.json FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum+TCC_MISS_sum -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_$tag/pmc.log 2>&1
rm -rf gpurun_out/prof_$tag/trace
echo done
