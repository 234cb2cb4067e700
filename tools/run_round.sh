# gpurun -- bash tools/run_round.sh <tag>: the whole GPU suite, then the profile set of tools/profile_round.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_$1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/prof_$1/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/prof_$1/gpu_tests.log
tail -3 gpurun_out/prof_$1/gpu_tests.log
bash tools/profile_round.sh $1
