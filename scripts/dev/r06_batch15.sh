#!/bin/bash
# K1 start stagger: sweep of the sleep unit on the kernel alone, then a same-box A/B of the whole forward:
# old kernel (HEAD~: no stagger code) vs new kernel with k1_phase = 0 / 4 / 8 / 16
export TMPDIR=/tmp
mkdir -p gpurun_out
# variant bits [19:12] = phase + 1:  off, 2, 4, 8, 16, 32
python scripts/dev/k1_q4.py c2 --q4 4096,12288,20480,36864,69632,135168 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_p_k1_phase_sweep.txt
tail -16 gpurun_out/r06_p_k1_phase_sweep.txt
export DMVS_ALLOW_DEV_BUILD=1
L=dmvsnet_amd/csrc/libdmvs_hip.so
bash scripts/dev/ab_bench.sh 5 old=dmvsnet_amd/csrc/dev/libdmvs_k1old.so off=$L:k1_phase=0 p4=$L:k1_phase=4 p8=$L:k1_phase=8 p16=$L:k1_phase=16 > gpurun_out/r06_p_ab_k1_phase.txt 2>&1
tail -8 gpurun_out/r06_p_ab_k1_phase.txt
