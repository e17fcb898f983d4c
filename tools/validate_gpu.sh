# full GPU suite + the default bench line (what the driver runs at round end)
cd /root/repo
mkdir -p gpurun_out/r06
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/gputest.log 2>&1
( time python bench.py ) > gpurun_out/r06/bench_default.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1
tail -n 6 gpurun_out/r06/gputest.log; tail -c 1500 gpurun_out/r06/bench_default.log; tail -n 3 gpurun_out/r06/smoke.log
