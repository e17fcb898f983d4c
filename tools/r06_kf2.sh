cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/kf_r.log; : > $out
for r in 4 8 2; do
  echo "== VORS_KF_R=$r" >> $out
  VORS_KF_R=$r MODES=c2f python tools/stage_times.py reference 4096 2>&1 | grep pairs >> $out
  VORS_KF_R=$r MODES=c2f python tools/stage_times.py fused 4096 2>&1 | grep pairs >> $out
done
export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pk
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pk -o p --output-format csv -- python /root/repo/bench.py --candidates c2f --arith reference --pairs 4096 --steps 2 --warmup 1 --no-secondary --no-pmc --no-sequences --cpu-pairs 0 --parity-pairs 0 > /dev/null 2>&1)
  f=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if any(s in k for s in ("keyframe_sparse", "rank_regions", "lm_ref_track", "pyramid_fused")):
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, v in agg.items():
    print(k, {c: f"{x:.3e}" for c, x in v.items()})
PY
done
cat $out
