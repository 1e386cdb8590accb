timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -12
for w in sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
echo "== $w"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_by_class'])"
done
