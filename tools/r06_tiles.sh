cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/rank_tiles.log; : > $out
python -m pytest tests/test_gpu_reference.py tests/test_gpu_trackers.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 >> $out
for rep in 1 2; do for rk in 1 2; do
  echo "== VORS_REF_RANK=$rk (rep $rep)" >> $out
  VORS_REF_RANK=$rk MODES=c2f python tools/stage_times.py reference 512 4096 2>&1 | grep pairs >> $out
done; done
cat $out
