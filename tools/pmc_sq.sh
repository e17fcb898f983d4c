#!/bin/bash
# SQ counters of the LM-stage kernels (two passes; run through gpurun). Development aid.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq; mkdir -p $OUT
ARGS="--pairs ${PAIRS:-2048} --steps 2 --warmup 1 --no-secondary --cpu-pairs 0"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o b -- python bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/p2 -o b -- python bench.py $ARGS > $OUT/p2.log 2>&1
python - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    fs=glob.glob(f"$OUT/{p}/**/*counter_collection.csv",recursive=True)
    if not fs: print(p,"no csv"); print(open(f"$OUT/{p}.log").read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]; k=k[k.find("vors::")+6:][:28] if "vors::" in k else k[:28]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items():
        if "lm_" in k: print(p,k,{a:f"{b:.4g}" for a,b in v.items()})
PY
rm -rf $OUT/p1 $OUT/p2
