cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/kf_pad_constl.log; : > $out
V=$PWD/visual-odometry-rs_amd/vors_amd
for rep in 1 2; do for tag in kp0l0 kp1l0 kp1l1; do
  echo "== $tag (rep $rep)" >> $out
  VORS_HIP_LIB=$V/libvors_hip_$tag.so MODES=c2f python tools/stage_times.py reference 512 4096 2>&1 | grep pairs >> $out
done; done
VORS_HIP_LIB=$V/libvors_hip_kp1l1.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_pyramid.py tests/test_gpu_trackers.py -x -q 2>&1 | tail -3 >> $out
cat $out
