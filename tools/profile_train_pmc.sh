# gpurun -- bash tools/profile_train_pmc.sh <tag>: SQ counters of the training step (config 3), two passes of <= 8 counters, kernels serialised
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_${1:-c3pmc}; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*LDS[A-Z0-9_]*\|SQ_[A-Z0-9_]*VMEM[A-Z0-9_]*" | sort -u > $O/counters_lds_vmem.txt
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_c3 -o p -- python3 bench.py --workload config3 --steps 2 --warmup 1 > /dev/null 2> $O/sq_c3.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/lds_c3 -o p -- python3 bench.py --workload config3 --steps 2 --warmup 1 > /dev/null 2> $O/lds_c3.err
if ! find $O/lds_c3 -name "*counter_collection.csv" | grep -q .; then
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds_c3 -o p -- python3 bench.py --workload config3 --steps 2 --warmup 1 > /dev/null 2>> $O/lds_c3.err
fi
python tools/summarize_train_pmc.py $O $O/pmc_sq_config3.csv | head -30
find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O; tail -3 $O/lds_c3.err
