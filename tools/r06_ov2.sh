cd /root/repo; mkdir -p gpurun_out/r06
out=gpurun_out/r06/overlap_sort.log; : > $out
for rep in 1 2; do for ov in 0 1; do
  echo "== VORS_OVERLAP_SORT=$ov (rep $rep)" >> $out
  VORS_OVERLAP_SORT=$ov MODES=c2f,dso python tools/stage_times.py reference 512 4096 2>&1 | grep pairs >> $out
done; done
python -m pytest tests/test_gpu_reference.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3 >> $out
cat $out
