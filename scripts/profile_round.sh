#!/bin/bash
# Round profile of one workload: rocprofv3 kernel-trace stats of the bench command, PMC passes (separate runs, counters never combined
# with traces), then the bench again so that its line picks up the PMC file.  usage (GPU box, repo root): scripts/profile_round.sh r03 [workload]
# The program stands directly behind `rocprofv3 ... --` (no env / bash -c hop), and the index is built BEFORE the traced run: the indexer is a
# child process with kernels of its own, and rocprofv3 writes one stats file per process — the summary kept is the one of the bench process
# (it holds k_ibf_query*), anything else is an error.
set -e
cd "$(dirname "$0")/.."
tag=${1:-rXX}
wl=${2:-config2}
export TMPDIR=/tmp
out=gpurun_out/prof_${tag}_${wl}
rm -rf $out
mkdir -p $out profiles
bench_args="--workload $wl --cpu-sample 0 --extras off"
python3 bench.py $bench_args --steps 1 --warmup 0 > /dev/null 2> $out/prepare.err          # builds genomes' index files if this box has none yet
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $bench_args --steps 5 --warmup 1 > $out/trace_bench.json 2> $out/trace.err
stats=""
for f in $(find $out/trace -name "*kernel_stats.csv"); do
    if grep -q "k_ibf_query" "$f"; then
        if [ -n "$stats" ]; then echo "profile_round: more than one process ran the sweep ($stats, $f)" >&2; exit 1; fi
        stats=$f
    fi
done
if [ -z "$stats" ]; then echo "profile_round: no kernel stats file holds k_ibf_query (wrong process traced?)" >&2; exit 1; fi
cp "$stats" profiles/${tag}_bench_${wl}_kernel_stats.csv
head -8 profiles/${tag}_bench_${wl}_kernel_stats.csv | cut -c1-160
python3 scripts/pmc.py $out/pmc.json FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum+TCC_MISS_sum,SQ_WAVES+SQ_INSTS_VALU+SQ_INSTS_SALU+SQ_INSTS_LDS+SQ_INSTS_VMEM_RD+SQ_WAVE_CYCLES+SQ_BUSY_CYCLES,TCC_EA0_RDREQ_sum+TCP_TCC_READ_REQ_sum -- python3 bench.py $bench_args --steps 2 --warmup 1 > $out/pmc.log 2>&1
grep -q "k_ibf_query" $out/pmc.json || { echo "profile_round: the PMC passes saw no sweep kernel" >&2; exit 1; }
cp $out/pmc.json profiles/${tag}_pmc_${wl}.json
python3 bench.py $bench_args --steps 10 --warmup 2 > $out/bench.json 2> $out/bench.err
cp $out/bench.json profiles/${tag}_bench_${wl}.json
tail -1 $out/bench.json | cut -c1-300
rm -rf $out/trace
echo done
