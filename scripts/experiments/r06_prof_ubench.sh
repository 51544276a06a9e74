cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_e; mkdir -p $OUT
B="$R/scripts/ubench/exact_mfma 10 2304 0"
rocprofv3 --kernel-include-regex "exact_rows" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4 -o p -- $B > /dev/null 2>&1
python $R/scripts/pmc_table.py "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" > $OUT/ub_pmc_stall.txt 2>&1
rocprofv3 --kernel-include-regex "exact_rows" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d /tmp/p5 -o p -- $B > /dev/null 2>&1
python $R/scripts/pmc_table.py "$(find /tmp/p5 -name '*counter_collection.csv' | head -1)" > $OUT/ub_pmc_mfma.txt 2>&1
rocprofv3 --kernel-include-regex "exact_rows" --pmc SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p6 -o p -- $B > /dev/null 2>&1
python $R/scripts/pmc_table.py "$(find /tmp/p6 -name '*counter_collection.csv' | head -1)" > $OUT/ub_pmc_misc.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -o p -- $B > /dev/null 2>&1
cp $(find /tmp/p7 -name '*kernel_stats.csv' | head -1) $OUT/ub_kernel_stats.csv
cat $OUT/ub_pmc_stall.txt $OUT/ub_pmc_mfma.txt $OUT/ub_pmc_misc.txt; head -5 $OUT/ub_kernel_stats.csv
