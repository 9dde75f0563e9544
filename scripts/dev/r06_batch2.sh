#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "coarse or bench_replicas or bench_hybrid or bench_eight" > $O/r06_b_pytest_subset.txt 2>&1
tail -5 $O/r06_b_pytest_subset.txt
python bench.py > $O/r06_b_bench_default.json 2> $O/r06_b_bench_default.err
python -c "import json;d=json.load(open('$O/r06_b_bench_default.json'));print(d['value'],d['ms_per_step'],d['legs_s'],d['legs_dropped']);print(d['roofline']);print(d['roofline_all']['prob_head']);print(d['roofline_all']['depth_regress'])"
