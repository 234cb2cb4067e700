cd $GRAFT_REPO_ROOT; O=gpurun_out/r4l; mkdir -p $O
python tools/dev_pro_err.py 2>&1 | tail -8
python -m pytest tests/test_gpu_blk.py -q -k "prologue or vs_oracle" 2>&1 | tail -3
