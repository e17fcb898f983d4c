# usage: bash tools/kernel_counters.sh <candidates> <arith> <outfile>   — SQ instruction-mix counters per kernel of 3 steps (1 warm-up + 2) at 4096 pairs
cd /root/repo; mkdir -p gpurun_out/r06
cand=$1; arith=$2; out=gpurun_out/r06/$3; : > $out
export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pk
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pk -o p --output-format csv -- python /root/repo/bench.py --candidates $cand --arith $arith --pairs 4096 --steps 2 --warmup 1 --no-secondary --no-pmc --no-sequences --cpu-pairs 0 --parity-pairs 0 > /dev/null 2>&1)
  f=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    if "synth" in k or "rocclr" in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k:70s}", {c: f"{x:.3e}" for c, x in v.items()})
PY
done
(cd /tmp && rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o p --output-format csv -- python /root/repo/bench.py --candidates $cand --arith $arith --pairs 4096 --steps 4 --warmup 1 --no-secondary --no-pmc --no-sequences --cpu-pairs 0 --parity-pairs 0 > /dev/null 2>&1)
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
python - "$f" >> $out <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    if "synth" in r["Name"]: continue
    print(f'{r["Name"].split("(")[0][:80]:80s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
cat $out
