#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "split_probe" 2>&1 | tail -15
python scripts/dev/split_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_i_split_probe.txt
cat gpurun_out/r06_i_split_probe.txt
