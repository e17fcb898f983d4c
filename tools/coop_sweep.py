"""Development aid (run through gpurun): the REFERENCE LM stage of small batches by wavefronts per pair of lm_ref_track_coop_kernel
(VORS_REF_COOP = 4 .. 8; 0 = one wavefront per pair) — where the sizing rule of lm_reference.hip refc_waves_per_pair comes from.
usage: python tools/coop_sweep.py [batch sizes ...]"""
import os, subprocess, sys
batches = sys.argv[1:] or ["256", "512", "768", "1024", "1536"]
for n in batches:
    for w in [int(x) for x in os.environ.get("COOPS", "0,4,5,6,7,8").split(",")]:
        env = dict(os.environ, VORS_REF_COOP=str(w), MODES=os.environ.get("MODES", "c2f,dso"))
        out = subprocess.run([sys.executable, "tools/stage_times.py", "reference", n], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if " pairs:" in line:
                print(f"coop {w}: {line}", flush=True)
