for sc in "" "--scene balls" "--scene bunny"; do
  for g in 1 16 24 32 40; do
    line=$(timeout -k 10 300 python bench.py --no-cpu-baseline $sc --tune 17=$g 2>/dev/null | tail -1)
    echo "$sc camgate $g $(echo "$line" | grep -o '"ms_per_step": [0-9.]*')"
  done
done
