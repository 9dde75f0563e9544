for r in ${ROWS:-0 8 16 24 32 48}; do echo "rows $r"; python scripts/layer_bench.py --only feat.conv0 --reps 50 --tune c8_rows=$r 2>&1 | grep "feat.conv0" | awk '{print $1, $4}'; done
