cd /root/repo
mkdir -p gpurun_out/r06
out=gpurun_out/r06/phases_fused.log; : > $out
for pairs in 1 512 4096; do for l in 5 3 0; do
  echo "== FUSED c2f+dso, $pairs pairs, level $l" >> $out
  PAIRS=$pairs MODES=0,2 ARITH=2 LEVEL=$l VLIB=libvors_hip_eph$l.so python tools/phase_times.py 2>&1 | grep mode >> $out
done; done
for pairs in 1 512 4096; do
  echo "== FUSED per-level, $pairs pairs" >> $out
  PAIRS=$pairs MODES=0,2 ARITH=2 VLIB=libvors_hip_eprof.so python tools/level_times.py 2>&1 | grep -v amdgpu >> $out
done
cat $out
