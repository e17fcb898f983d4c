"""Merge gpurun_out/prof_<tag>/lm_counters_<tag>.json (written by tools/profile.sh on the GPU box) into profiles/lm_counters.json under
the workload key bench.py looks up, and copy the summary to profiles/.   usage: python tools/make_lm_counters.py TAG KEY"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
entry = json.load(open(os.path.join(src, f"lm_counters_{tag}.json")))
path = os.path.join(ROOT, "profiles", "lm_counters.json")
table = json.load(open(path)) if os.path.exists(path) else {
    "_comment": "Counters of the LM stage of ONE bench step per workload, from rocprofv3 passes of `bench.py <workload> --no-secondary --cpu-pairs 0` "
                "(tools/profile.sh: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_*; each its own run). traffic_bytes = "
                "FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB. SQ_ACTIVE_*/SQ_WAVE_CYCLES/SQ_WAIT_* are quad-cycles. "
                "bench.py copies the entry of the workload it runs into roofline.traffic / valu_issue_frac (labelled from_profile). "
                "Key = <candidates>_<arith>_<cols>x<rows>_L<levels>_<pairs>pairs[_huber<d>]."}
table[key] = entry
json.dump(table, open(path, "w"), indent=1)
shutil.copy(os.path.join(src, f"summary_{tag}.md"), os.path.join(ROOT, "profiles", f"{tag}_summary.md"))
print(key, entry)
