#!/bin/bash
# Evidence set of one round on the GPU box (run from the repo root through gpurun):
#   scripts/gpu_profile.sh TAG [bench|stats|pmc|layers ...]     (default: all four)
# writes gpurun_out/TAG_*; copy what should be judged into profiles/.
set -u
TAG=${1:?tag}; shift
WHAT=${*:-bench stats pmc layers k1 configs}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
for w in $WHAT; do
case $w in
bench)
    python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err" ;;
stats)
    for mode in default single_stream; do
        flag=""; [ $mode = single_stream ] && flag="--single-stream"
        rm -rf /tmp/prof_$mode
        (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o p -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-aten-gpu-baseline --no-live-traffic --no-full-outputs $flag > /tmp/prof_$mode.log 2>&1)
        db=$(find /tmp/prof_$mode -name '*.db' | head -1)
        if [ -n "$db" ]; then python scripts/rocpd_summary.py "$db" "$OUT/${TAG}_rocprof_kernel_stats_$mode.md" > /dev/null
        else
            csv=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1)
            [ -n "$csv" ] && cp "$csv" "$OUT/${TAG}_rocprof_kernel_stats_$mode.csv"
        fi
    done ;;
pmc)
    : > "$OUT/${TAG}_pmc_fetch_write.txt"
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-aten-gpu-baseline --no-full-outputs --no-kernel-timing --single-stream --launch-log /tmp/launch_$c.json > /tmp/pmc_$c.log 2>&1)
        csv=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
        [ -n "$csv" ] && python scripts/pmc_summary.py --launch-log /tmp/launch_$c.json "$csv" >> "$OUT/${TAG}_pmc_fetch_write.txt"
    done ;;
layers)
    python scripts/layer_bench.py > "$OUT/${TAG}_layer_table.md" 2> "$OUT/${TAG}_layer_table.err" ;;
k1)
    # K1 alone: smooth planes (scripts/k1_bench.py), the real bench inputs incl. the generic kernel, SQ / TCC counters
    python scripts/k1_bench.py > "$OUT/${TAG}_k1_smooth.json" 2> /dev/null
    python scripts/dev/k1_q4.py c2 --q4 0,8,16 --hwc 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_k1_ab.txt"
    bash scripts/dev/k1_q4_pmc.sh "$TAG" 0 both > /dev/null 2>&1 ;;
configs)
    for c in c3 c4 c5 dtu tnt; do python bench.py --config $c --steps 10 --warmup 3 --no-aten-gpu-baseline --no-live-traffic > "$OUT/${TAG}_bench_$c.json" 2> /dev/null; done
    python bench.py --config c5 --steps 10 --warmup 3 --feature-dtype f16 --no-cpu-baseline --no-aten-gpu-baseline --no-live-traffic > "$OUT/${TAG}_bench_c5_f16_features.json" 2> /dev/null ;;
esac
done
ls -la "$OUT" | grep "${TAG}_"
