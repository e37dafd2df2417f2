#!/bin/bash
# Round profile: the default bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (separate runs, counters never combined
# with traces), then the bench again so that its line picks up the PMC file.  usage (GPU box, repo root): scripts/profile_round.sh r02 [workload]
set -e
cd "$(dirname "$0")/.."
tag=${1:-rXX}
wl=${2:-config2}
export TMPDIR=/tmp
out=gpurun_out/prof_${tag}_${wl}
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 > $out/trace_bench.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
head -8 $out/kernel_stats.csv | cut -c1-160
python3 scripts/pmc.py $out/pmc.json FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum+TCC_MISS_sum,SQ_WAVES+SQ_INSTS_VALU+SQ_INSTS_SALU+SQ_INSTS_LDS+SQ_INSTS_VMEM_RD+SQ_WAVE_CYCLES+SQ_BUSY_CYCLES,TCC_EA0_RDREQ_sum+TCP_TCC_READ_REQ_sum -- python3 bench.py --workload $wl --steps 2 --warmup 1 --cpu-sample 0 > $out/pmc.log 2>&1
mkdir -p profiles
cp $out/pmc.json profiles/${tag}_pmc_${wl}.json
cp $out/kernel_stats.csv profiles/${tag}_bench_${wl}_kernel_stats.csv
python3 bench.py --workload $wl --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err
cp $out/bench.json profiles/${tag}_bench_${wl}.json
tail -1 $out/bench.json | cut -c1-400
rm -rf $out/trace
echo done
