#!/bin/bash
# the round's evidence set (scripts/gpu_profile.sh) + the default bench line as the driver runs it
export TMPDIR=/tmp
bash scripts/gpu_profile.sh r06_k bench stats pmc layers k1 > gpurun_out/r06_k_profile.log 2>&1
tail -3 gpurun_out/r06_k_profile.log
python -c "
import json
d=json.load(open('gpurun_out/r06_k_bench_default.json'))
print(d['value'], d['ms_per_step'], d['ms_per_stage'])
print(d['roofline'])
print(d['legs_s'])
print(d['parity']['depth_rel_l1'], d['warp_hbm_frac'], d.get('warp_hbm_frac_coherent'))
"
