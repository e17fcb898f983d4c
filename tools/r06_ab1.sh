# round 6 A/B: small-batch overlap of the current pyramid and the parked ahead-step of the workgroup-per-pair REFERENCE kernel
cd /root/repo
mkdir -p gpurun_out/r06
python -c "import torch; p=torch.cuda.get_device_properties(0); print('shared_memory_per_block', p.shared_memory_per_block, getattr(p,'shared_memory_per_multiprocessor',None), p.multi_processor_count)" > gpurun_out/r06/ab1.log 2>&1
for ov in 0 1024; do for ah in 0 2; do
  echo "== VORS_OVERLAP_MAX_PAIRS=$ov VORS_REF_AHEAD=$ah" >> gpurun_out/r06/ab1.log
  VORS_OVERLAP_MAX_PAIRS=$ov VORS_REF_AHEAD=$ah MODES=c2f,dso python tools/stage_times.py reference 512 >> gpurun_out/r06/ab1.log 2>&1
done; done
for ov in 0 1024; do
  echo "== fused VORS_OVERLAP_MAX_PAIRS=$ov" >> gpurun_out/r06/ab1.log
  VORS_OVERLAP_MAX_PAIRS=$ov python tools/stage_times.py fused 512 >> gpurun_out/r06/ab1.log 2>&1
done
echo "== single tracker latency (default knobs)" >> gpurun_out/r06/ab1.log
python tools/seq_latency.py 2>&1 | grep "rep 2" >> gpurun_out/r06/ab1.log
echo "== single tracker, VORS_REF_AHEAD=2 forced (8-wave form parked)" >> gpurun_out/r06/ab1.log
VORS_REF_AHEAD=2 python tools/seq_latency.py 2>&1 | grep "rep 2" >> gpurun_out/r06/ab1.log
python -m pytest tests/test_gpu_reference.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5 >> gpurun_out/r06/ab1.log
cat gpurun_out/r06/ab1.log
