#!/bin/bash
# full GPU suite + the evidence set of the round's FINAL build (scripts/gpu_profile.sh r06_w)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r06_w_pytest_gpu_full.txt 2>&1
tail -2 gpurun_out/r06_w_pytest_gpu_full.txt
bash scripts/gpu_profile.sh r06_w bench stats pmc layers k1 > gpurun_out/r06_w_profile.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r06_w_bench_default.json'))
print(d['value'], d['ms_per_step'], d['ms_per_stage'], d.get('value_full_outputs',{}).get('value'))
print({k:(round(v['ms_per_map'],3), round(v.get('frac',0),3)) for k,v in d['roofline_all'].items()})
print(d['roofline']['frac'], d['roofline']['executed_frac'], d['roofline']['largest_kernel']['frac'], d['warp_hbm_frac'], d.get('warp_hbm_frac_coherent'))
"
