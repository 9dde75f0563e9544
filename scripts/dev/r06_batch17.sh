#!/bin/bash
# `prob` -> K4 (dmvs_prob_regress + dmvs_depth_select): parity, then a same-box A/B of the whole forward against the two kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prob_regress or fused_heads or depth_regress or end_to_end" 2>&1 | tail -5
L=dmvsnet_amd/csrc/libdmvs_hip.so
bash scripts/dev/ab_bench.sh 6 fused=$L two=$L@--no-prob-fused > gpurun_out/r06_s_ab_prob_fused.txt 2>&1
tail -4 gpurun_out/r06_s_ab_prob_fused.txt
python bench.py --no-cpu-baseline --no-aten-gpu-baseline --no-live-traffic --no-full-outputs > gpurun_out/r06_s_bench_fused.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aten-gpu-baseline --no-live-traffic --no-full-outputs --no-prob-fused > gpurun_out/r06_s_bench_two.json 2>/dev/null
python - <<'PY'
import json
for n in ("fused", "two"):
    d = json.load(open(f"gpurun_out/r06_s_bench_{n}.json"))
    ra = d["roofline_all"]
    print(n, round(d["value"], 2), {k: round(v, 3) for k, v in d["ms_per_stage"].items()},
          "prob", round(ra["prob_head"]["ms_per_map"], 3), "k4", round(ra["depth_regress"]["ms_per_map"], 3))
PY
