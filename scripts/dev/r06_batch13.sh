#!/bin/bash
# full GPU suite + the round's evidence set (scripts/gpu_profile.sh) on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r06_n_pytest_gpu_full.txt 2>&1
tail -2 gpurun_out/r06_n_pytest_gpu_full.txt
bash scripts/gpu_profile.sh r06_n bench stats pmc layers k1 > gpurun_out/r06_n_profile.log 2>&1
tail -3 gpurun_out/r06_n_profile.log
python -c "
import json
d=json.load(open('gpurun_out/r06_n_bench_default.json'))
print(d['value'], d['ms_per_step'], d['ms_per_stage'])
print(d['roofline'])
print(d['legs_s'])
print(d['parity']['depth_rel_l1'], d['warp_hbm_frac'], d.get('warp_hbm_frac_coherent'))
"
