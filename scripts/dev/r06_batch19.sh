#!/bin/bash
# K1 start stagger, order-balanced: the first variant timed in a pass runs slower whatever it is (clocks / caches settle), which the
# earlier sweeps did not control for.  ABBA BAAB order, on = 36864 (8), off = 4096; totals are sums over the 4 appearances.
export TMPDIR=/tmp
python scripts/dev/k1_q4.py c2 --q4 36864,4096,4096,36864,4096,36864,36864,4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_u_k1_stagger_balanced.txt
