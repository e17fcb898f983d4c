cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/rank_tiles_kernels.log; : > $out
export TMPDIR=/tmp
for rk in 1 2; do
(cd /tmp && rm -rf /tmp/pk && VORS_REF_RANK=$rk rocprofv3 --kernel-trace --stats -d /tmp/pk -o p --output-format csv -- python /root/repo/bench.py --candidates c2f --arith reference --pairs 4096 --steps 4 --warmup 1 --no-secondary --no-pmc --no-sequences --cpu-pairs 0 --parity-pairs 0 > /dev/null 2>&1)
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
echo "== VORS_REF_RANK=$rk" >> $out
python - "$f" >> $out <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    if "synth" in r["Name"]: continue
    print(f'{r["Name"].split("(")[0][:80]:80s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
done
cat $out
