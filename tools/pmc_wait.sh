#!/bin/bash
# Where do the wavefronts of the per-pair LM kernels wait?  Three rocprofv3 --pmc passes (kernel trace only) of a bench.py workload, LM-stage
# kernels summed per counter, plus derived ratios (average LDS / VMEM instruction latency = level sum / instruction count). Development aid,
# run through gpurun:   [KERNELS='regex of kernel names'] bash tools/pmc_wait.sh TAG --candidates c2f --arith reference
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wait_$TAG; mkdir -p $OUT
ARGS="$@ --steps 2 --warmup 1 --no-secondary --cpu-pairs 0 --parity-pairs 0 --no-pmc --no-sequences"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -o b -- python bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VALU --output-format csv -d $OUT/p2 -o b -- python bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT --output-format csv -d $OUT/p3 -o b -- python bench.py $ARGS > $OUT/p3.log 2>&1
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); names=set()
for p in ("p1","p2","p3"):
    fs=glob.glob(f"$OUT/{p}/**/*counter_collection.csv",recursive=True)
    if not fs: print(p,"no csv"); print(open(f"$OUT/{p}.log").read()[-400:]); continue
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        import os, re
        pat = os.environ.get("KERNELS", "lm_track_kernel|lm_ref_track_kernel|lm_split")
        if re.search(pat, k):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); names.add(k[k.find("vors::"):][:60])
print("$TAG kernels:", sorted(names))
for k in sorted(tot): print(f"  {k:24s} {tot[k]:.4g}")
g=lambda k: tot.get(k,0.0)
if g("SQ_WAVE_CYCLES"):
    print(f"  wait_any / wave_cycles {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.3f}; wait_inst_any / wave_cycles {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f}; wait_inst_lds / wave_cycles {g('SQ_WAIT_INST_LDS')/g('SQ_WAVE_CYCLES'):.3f}")
    print(f"  active: any {g('SQ_ACTIVE_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f} valu {g('SQ_ACTIVE_INST_VALU')/g('SQ_WAVE_CYCLES'):.3f} lds {g('SQ_ACTIVE_INST_LDS')/g('SQ_WAVE_CYCLES'):.3f} sca {g('SQ_ACTIVE_INST_SCA')/g('SQ_WAVE_CYCLES'):.3f} vmem {g('SQ_ACTIVE_INST_VMEM')/g('SQ_WAVE_CYCLES'):.3f} (of wave-cycles)")
    print(f"  waves resident per SIMD-busy cycle {4*g('SQ_WAVE_CYCLES')/max(g('SQ_BUSY_CYCLES'),1):.2f} (x4: quad-cycle units?)")
if g("SQ_INSTS_LDS"): print(f"  avg LDS instruction latency {g('SQ_INST_LEVEL_LDS')/g('SQ_INSTS_LDS'):.1f} cycles; bank-conflict cycles / lds active {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):.3f}")
if g("SQ_INSTS_VMEM_RD"): print(f"  avg VMEM instruction latency {g('SQ_INST_LEVEL_VMEM')/g('SQ_INSTS_VMEM_RD'):.1f} cycles")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
