cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/r06_stage_times_512_vs_4096.log
echo "# tools/stage_times.py fused, then reference: 512 vs 4096 pairs, SINGLE STREAM (round 6 final build)" > $out
python tools/stage_times.py fused 512 4096 2>&1 | grep -E "pairs" >> $out
python tools/stage_times.py reference 512 4096 2>&1 | grep -E "pairs" >> $out
echo "# tools/pipeline_small_batches.py: the same steps as a continuous feed through vors_pipeline_* (rings of 2 and 3 batch handles on internal streams)" >> $out
python tools/pipeline_small_batches.py 2>&1 | grep -v amdgpu >> $out
echo "# tools/ring_depth_sweep.py: ring depth 1 / 2 / 3 / 4 / 6, 100 steps of 512 pairs, 40 of 4096" >> $out
python tools/ring_depth_sweep.py 2>&1 | grep -v amdgpu >> $out
cat $out
