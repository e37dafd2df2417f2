#!/bin/bash
# builds libdyhip.so with different K3 occupancy targets on the GPU box and runs the bench for each (tuning experiment)
cd "$(dirname "$0")/.."
for w in 4 5 6; do
  DY_HIPCC_FLAGS="-DDY_K3_WAVES=$w" python3 -m dream_yara_amd.build --force > gpurun_out/tune_build_$w.log 2>&1 || { echo "build failed for $w"; tail -5 gpurun_out/tune_build_$w.log; exit 1; }
  timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 8 > gpurun_out/tune_k3_$w.log 2>&1
  tail -1 gpurun_out/tune_k3_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w', round(d['value']/1e6,1), 'Mreads/s', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['config']['stage_ms_per_step'].items()})"
done
