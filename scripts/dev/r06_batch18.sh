#!/bin/bash
# full GPU suite + the evidence set of the round's final build (scripts/gpu_profile.sh r06_t, all parts)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r06_t_pytest_gpu_full.txt 2>&1
tail -2 gpurun_out/r06_t_pytest_gpu_full.txt
bash scripts/gpu_profile.sh r06_t bench stats pmc layers k1 configs > gpurun_out/r06_t_profile.log 2>&1
tail -3 gpurun_out/r06_t_profile.log
python -c "
import json
d=json.load(open('gpurun_out/r06_t_bench_default.json'))
print(d['value'], d['ms_per_step'], d['ms_per_stage'], d.get('value_full_outputs',{}).get('value'))
print(d['roofline'])
print({k:(round(v['ms_per_map'],3), round(v.get('frac',0),3)) for k,v in d['roofline_all'].items()})
print(d['legs_s'])
print(d['parity']['depth_rel_l1'], d['warp_hbm_frac'], d.get('warp_hbm_frac_coherent'))
"
