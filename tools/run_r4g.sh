cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_configs.py -x -q -s -k "full_size_bf16_gradients or config5_like" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; grep -v "^$" $O/t1.log | tail -12
timeout 900 python -m pytest tests/test_gpu_train_fused.py -x -q -k "side_stream" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -5 $O/t2.log
