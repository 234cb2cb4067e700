# gpurun -- bash tools/profile_wgrad_blk.sh: SQ counters of the staged-unit weight gradient against the dense-over-taps kernel (tools/dev_wgrad_blk.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_wgrad_blk; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_c3 -o p -- python3 tools/dev_wgrad_blk.py > /dev/null 2> $O/sq.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/lds_c3 -o p -- python3 tools/dev_wgrad_blk.py > /dev/null 2> $O/lds.err
python tools/summarize_train_pmc.py $O $O/pmc_wgrad_blk.csv | grep "kernel\|wgrad"
find $O -name "*counter_collection.csv" -delete
