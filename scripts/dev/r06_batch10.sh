#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r06_j_pytest_gpu_full.txt 2>&1
tail -8 gpurun_out/r06_j_pytest_gpu_full.txt
