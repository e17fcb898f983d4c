cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/kf_r_constl.log; : > $out
for rep in 1 2; do for r in 4 8 2; do
  echo "== VORS_KF_R=$r (rep $rep)" >> $out
  VORS_KF_R=$r MODES=c2f python tools/stage_times.py reference 4096 2>&1 | grep pairs >> $out
done; done
MODES=c2f python tools/stage_times.py fused 512 4096 2>&1 | grep pairs >> $out
cat $out
