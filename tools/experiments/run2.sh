set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
./tools/experiments/valu_lds_rate 2>&1 | tee $O/valu_lds_rate.txt
cd /tmp && export TMPDIR=/tmp
for q in 1 0; do
export MVDETR_MSDA_QUAD=$q
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_a$q -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE -d $O/pmc_b$q -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL -d $O/pmc_c$q -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/pmc_a$q/p_results.db $O/pmc_b$q/p_results.db $O/pmc_c$q/p_results.db --filter msda_fwd > $O/pmc_quad$q.txt
rm -rf $O/pmc_a$q $O/pmc_b$q $O/pmc_c$q
done
cat $O/pmc_quad1.txt $O/pmc_quad0.txt
