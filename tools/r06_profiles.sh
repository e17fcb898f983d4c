# round 6: the committed rocprofv3 summaries of the final build (tools/profile.sh: kernel trace + FETCH / WRITE / SQ passes, each its own run)
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile.sh r06_dense_4096pairs_fused --steps 5 --warmup 2
bash tools/profile.sh r06_dense_4096pairs_reference --arith reference --steps 3 --warmup 1
bash tools/profile.sh r06_c2f_4096pairs_fused --candidates c2f --steps 5 --warmup 2
bash tools/profile.sh r06_c2f_4096pairs_reference --candidates c2f --arith reference --steps 5 --warmup 2
bash tools/profile.sh r06_dso_4096pairs_fused --candidates dso --steps 5 --warmup 2
bash tools/profile.sh r06_dso_4096pairs_reference --candidates dso --arith reference --steps 5 --warmup 2
bash tools/profile.sh r06_config5_1280x960_512pairs_fused --pairs 512 --rows 960 --cols 1280 --levels 7 --huber 10 --steps 5 --warmup 2
bash tools/profile.sh r06_config5_1280x960_512pairs_reference --pairs 512 --rows 960 --cols 1280 --levels 7 --huber 10 --arith reference --steps 3 --warmup 1
ls gpurun_out/prof_r06_*/
