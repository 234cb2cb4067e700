# Collect the per-round profile set on the GPU box (run through gpurun: `gpurun -- bash tools/profile_round.sh <tag>`), then
# `python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<tag>` here.  Two un-profiled default lines (four tiles in
# flight = the headline configuration; one tile at a time), kernel-trace stats for BOTH, and the three PMC passes (one tile at a
# time: cleaner per-kernel attribution; counters never combined with trace domains other than the kernel trace).
export GPU_MAX_HW_QUEUES=8   # rocprofv3 starts the HIP runtime before python runs: the package's own default would come too late
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_${1:-r2}; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
python bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
python bench.py --tiles-in-flight 1 $Q > $O/bench_unprofiled_1_in_flight.json 2>> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 6 --warmup 2 $Q > $O/bench_line.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o p -- python3 bench.py --steps 6 --warmup 2 --tiles-in-flight 1 $Q > $O/bench_line_1_in_flight.json 2> $O/stats1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --tiles-in-flight 1 $Q > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 bench.py --steps 3 --warmup 1 --tiles-in-flight 1 $Q > /dev/null 2> $O/write.err
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o p -- python3 bench.py --steps 3 --warmup 1 --tiles-in-flight 1 $Q > /dev/null 2> $O/sq.err
python tools/power_kernels.py > $O/power_per_layer.txt 2> $O/power.err
python tools/dev_train_layers.py > $O/train_per_layer.txt 2> $O/train_layers.err
python bench.py --workload config3 --steps 4 --warmup 2 > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --workload config5 $Q > $O/bench_config5.json 2> $O/bench_config5.err
python bench.py --workload config4 $Q > $O/bench_config4.json 2> $O/bench_config4.err
mkdir -p $O/c3; rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o p -- python3 bench.py --workload config3 --steps 4 --warmup 2 > $O/bench_config3_profiled.json 2> $O/c3.err
python tools/trace_gaps.py $O/c3 > $O/config3_timeline.txt 2>&1
find $O/c3 -name "*kernel_trace.csv" -delete
python tools/dev_train_host.py > $O/config3_phases.txt 2>&1
TL_BENCH_ADAMW=foreach python tools/dev_train_host.py >> $O/config3_phases.txt 2>&1
python tools/dev_wgrad_dense.py > $O/wgrad_dense_vs_pair_list.txt 2>&1
python tools/dev_conv_table.py > $O/conv_launch_table.txt 2>&1
tail -c 400 $O/bench_unprofiled.json; ls $O/*
# round 5: host cost of a lone forward through tl_forward / the Python-driven engine, the timeline of one lone forward, the parity modes per layer
python tools/dev_fwd_host.py > $O/fwd_host_tl_forward.txt 2>&1
TL_EXEC=0 python tools/dev_fwd_host.py > $O/fwd_host_python_engine.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr1 -o p -- python3 bench.py --steps 6 --warmup 2 --tiles-in-flight 1 $Q > /dev/null 2> $O/tr1.err
python tools/trace_forward.py $O/tr1 3 > $O/forward_timeline_lone.txt 2>&1
find $O/tr1 -name "*kernel_trace.csv" -delete
python bench.py --dtype bf16x3 --no-cpu-baseline --no-power-probe --no-extra-workloads --tiles-in-flight 1 --steps 5 --warmup 2 --layer-table $O/layer_table_bf16x3.txt > $O/bench_bf16x3.json 2>> $O/bench_unprofiled.err
python bench.py --dtype fp32 --no-cpu-baseline --no-power-probe --no-extra-workloads --tiles-in-flight 1 --steps 5 --warmup 2 --layer-table $O/layer_table_fp32.txt > $O/bench_fp32.json 2>> $O/bench_unprofiled.err
python bench.py --tiles-in-flight 1 $Q --layer-table $O/layer_table_bf16.txt > /dev/null 2>> $O/bench_unprofiled.err
python tools/dev_k8.py > $O/k8_table.txt 2>&1
python tools/dev_geom_value.py > $O/geom_value.txt 2>&1
# round 6: the LDS scatter-add roof, the parity-fast mode's staged level-1 kernel and its level-2 / 3 quad-gather kernels
[ -x tools/lds_add_roof ] && ./tools/lds_add_roof > $O/lds_add_roof.txt 2>&1
python tools/dev_x3_blk.py > $O/x3_blk.txt 2>&1
python tools/dev_x3_l2.py > $O/x3_l2.txt 2>&1
