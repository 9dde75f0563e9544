#!/bin/bash
# same-box A/B of whole-forward throughput between library builds (box-to-box spread is +-1.5 %, the changes being compared
# are often smaller): alternates `bench.py` runs, prints depth-maps/s per run and the medians.
#   scripts/dev/ab_bench.sh ROUNDS name1=path1.so name2=path2.so[:knob=value][@--flag,--flag] ... [-- extra bench.py args]
rounds=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq $rounds); do
  for l in "${libs[@]}"; do
    name=${l%%=*}; path=${l#*=}; extra=""
    case "$path" in *@*) extra="${path#*@}"; path=${path%%@*};; esac            # name=lib.so@--bench-flag (spaces as ,)
    extra=${extra//,/ }
    case "$path" in *:*) extra="$extra --tune ${path#*:}"; path=${path%%:*};; esac     # name=lib.so:knob=value
    v=$(DMVS_LIB=$path python bench.py $extra --no-cpu-baseline --no-aten-gpu-baseline --no-full-outputs --no-kernel-timing --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],2))")
    echo "round $r $name $v"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    _, _, n, v = l.split()
    d[n].append(float(v))
for n, v in d.items():
    print(f"{n:12s} median {statistics.median(v):.2f}  min {min(v):.2f}  max {max(v):.2f}  n={len(v)}")
PY
