#!/bin/bash
# K1 experiments: plane chunks walked by one workgroup (variant bits [10:8]) and a start stagger of the
# first workgroup generation (bits [19:12]) against the default launch, smooth planes + the bench's real inputs
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/dev/k1_q4.py c2 --q4 0,512,1024,16384,32768,17408 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_o_k1_chunk_loop_phase.txt
cat gpurun_out/r06_o_k1_chunk_loop_phase.txt | tail -20
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warp_corr" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "bench" 2>&1 | tail -5
