#!/bin/bash
# PMC counters for the LM kernel (own run, --kernel-trace only). usage: tools/pmc.sh TAG "COUNTERS" [bench args]
TAG=$1; CNT=$2; shift; shift
ARGS=${@:-"--steps 3 --warmup 1 --no-secondary --cpu-pairs 0"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o run -- python bench.py $ARGS > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "vors" not in k: continue
    print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
tail -1 $OUT/log.txt | cut -c1-300
