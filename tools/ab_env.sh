#!/bin/bash
# Development aid (run through gpurun): A/B of one environment knob on the bench step.  usage: tools/ab_env.sh VAR "dense c2f dso" [VALUE ...]
var=$1; modes=${2:-dense}; shift; shift; vals=${@:-1}
for mode in $modes; do for f in "" $vals "" $vals; do
env ${f:+$var=$f} python bench.py --candidates $mode --no-pmc --no-sequences --no-secondary --parity-pairs 0 --cpu-pairs 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode $var=$f', round(d['value']), d['ms_per_step'], d.get('stages_ms'))"
done; done
