cd /root/repo
mkdir -p gpurun_out/r06
python tools/stage_times.py fused 512 4096 > gpurun_out/r06/base_stage_fused.log 2>&1
python tools/stage_times.py reference 512 4096 > gpurun_out/r06/base_stage_reference.log 2>&1
python tools/seq_latency.py > gpurun_out/r06/base_seq_latency.log 2>&1
L=$PWD/visual-odometry-rs_amd/vors_amd/libvors_hip_rtiming.so
for n in 1 64 512; do VORS_HIP_LIB=$L python tools/ref_profile_coop.py $n c2f dso >> gpurun_out/r06/base_coop_profile.log 2>&1; done
tail -n 30 gpurun_out/r06/*.log
