for mode in dense c2f dso; do for f in 1 0 1 0; do
VORS_PYRAMID_FUSED=$f python bench.py --candidates $mode --no-pmc --no-sequences --no-secondary --parity-pairs 0 --cpu-pairs 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode fused=$f', round(d['value']), d['ms_per_step'], d.get('stages_ms'))"
done; done
