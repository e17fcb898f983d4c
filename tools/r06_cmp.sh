cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/compact_batched.log; : > $out
MODES=c2f python tools/stage_times.py fused 512 4096 2>&1 | grep pairs >> $out
MODES=c2f python tools/stage_times.py exact 4096 2>&1 | grep pairs >> $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_trackers.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 >> $out
cat $out
