#!/bin/bash
# Development aid (run through gpurun): A/B of the DSO keyframe-stage kernels at 4096 pairs with an environment switch, e.g.
#   bash tools/ab_dso.sh VORS_DSO_FIRST_MAXIMA 1 0        -> median duration per kernel over the 4096-pair launches of tools/stage_times.py
VAR=${1:-VORS_DSO_FIRST_MAXIMA}; shift; VALS=${@:-1 0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$v -o b -- python tools/stage_times.py ${ARITH:-fused} > /dev/null 2>&1
  f=$(find gpurun_out/ab_$v -name "*kernel_trace.csv" | head -1)
  echo "== $VAR=$v"; python - "$f" <<'PY'
import csv,sys,collections,statistics
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'dso_' in n or 'mask_sparse' in n or 'sort_colmajor' in n or 'sparse_fill' in n:
        wg=int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z'])
        d[(n[:48],wg)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
big={}
for (n,wg),v in d.items():
    if n not in big or wg>big[n][0]: big[n]=(wg,v)
for n,(wg,v) in sorted(big.items()): print(f"  {n:48s} {len(v):3d} launches of the largest grid: median {statistics.median(v):8.1f} us  min {min(v):8.1f}")
PY
  rm -rf gpurun_out/ab_$v
done; done
