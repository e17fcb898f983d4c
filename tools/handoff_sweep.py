"""Development aid (run through gpurun): the REFERENCE LM stage of the 4096-pair batch by hand-over point (VORS_REF_HANDOFF = percent of the
pairs finished before the stragglers move to workgroups) and workgroup size of the finishing launch (VORS_REF_HANDOFF_WAVES).
usage: [MODES=c2f,dso,dense] python tools/handoff_sweep.py"""
import os, subprocess, sys
for pct in os.environ.get("PCTS", "0,50,60,65,70,80").split(","):
    for hw in os.environ.get("HWS", "3,4,5").split(","):
        if pct == "0" and hw != "4":
            continue
        env = dict(os.environ, VORS_REF_HANDOFF=pct, VORS_REF_HANDOFF_WAVES=hw, MODES=os.environ.get("MODES", "c2f,dso"))
        out = subprocess.run([sys.executable, "tools/stage_times.py", "reference", "4096"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if " pairs:" in line:
                print(f"hand-over {pct:>2s} % -> {hw} wavefronts: {line}", flush=True)
