# GPU box: headline rate vs number of detector streams (same box, back to back)
for s in 1 2 3 4; do
  python bench.py --no-extra --no-cpu-baseline --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', $s, d['value'], d['ms_per_step'])"
done
