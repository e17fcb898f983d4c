#!/bin/bash
# Development aid (run through gpurun): LM-stage kernel time of the last bench step for several builds of the library
# (tools/build_variant.sh TAG ... -> libvors_hip_eTAG.so).   usage: tools/lm_variants.sh TAG [TAG ...]   (env PAIRS, default 4096)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  lib=$GRAFT_REPO_ROOT/visual-odometry-rs_amd/vors_amd/libvors_hip_e$tag.so
  [ "$tag" = "base" ] && lib=$GRAFT_REPO_ROOT/visual-odometry-rs_amd/vors_amd/libvors_hip.so
  out=gpurun_out/var_$tag; rm -rf $out; mkdir -p $out
  VORS_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python bench.py --pairs ${PAIRS:-4096} --steps 3 --warmup 1 --no-secondary --cpu-pairs 0 > $out/bench.log 2>&1
  python - "$tag" "$out" <<'PY'
import csv, glob, sys, json
tag, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(out + "/bench.log").read().strip().splitlines()[-1])
    head = f"{d['value']:.0f} pairs/s, lm {d['stages_ms']['lm']:.2f} ms, failed {d.get('failed_pairs')}, gt-err med {d['pose_err_vs_ground_truth']['median']:.5f}"
except Exception as e:
    head = "bench failed: " + open(out + "/bench.log").read()[-300:]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
if not f:
    print(tag, head, "no trace"); sys.exit()
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lm_track_kernel" in r["Kernel_Name"]]
rows = rows[idx[-3]:]
tot = {}
for r in rows:
    n = r["Kernel_Name"]
    k = "full" if "lm_split_eval_kernel<false, false" in n or "lm_split_eval_kernel<true, false" in n else "energy" if "lm_split_eval_kernel" in n else "step" if "step" in n else "track" if "lm_track" in n else "other"
    tot.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
s = "; ".join(f"{k} {sum(v)/1e3:.2f} ms (top {', '.join(f'{x:.0f}' for x in sorted(v, reverse=True)[:4])} us)" for k, v in tot.items())
print(f"[{tag}] {head}\n      {s}")
PY
  rm -rf $out/trace
done
