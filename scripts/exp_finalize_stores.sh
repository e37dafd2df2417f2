#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for f in 0 4 8 16 28; do
  DY_DEBUG_FINALIZE=$f rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/prof_tmp.log 2>&1
  find gpurun_out/prof_tmp -name "*kernel_stats.csv" -exec cp {} gpurun_out/ks_tmp.csv \;
  rm -rf gpurun_out/prof_tmp
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/ks_tmp.csv')):
    n=r['Name'].split('(')[0]
    if n.endswith('k_finalize_simple'): print('flags=$f k_finalize_simple %.1f us' % (float(r['AverageNs'])/1e3))
"
done
