#!/bin/bash
# SQ counters of the exact stage's DDC (compile-time form) alone
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_qd; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --no-ab --no-host-fed --sync"
rocprofv3 --kernel-include-regex "verify_ddc|window_kernel" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3 -o p -- $PM > /dev/null 2> $O/err
python $R/scripts/pmc_table.py "$(find /tmp/p3 -name '*counter_collection.csv' | head -1)" > $O/pmc_sq.txt 2>> $O/err
rocprofv3 --kernel-include-regex "verify_ddc|window_kernel" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4 -o p -- $PM > /dev/null 2>> $O/err
python $R/scripts/pmc_table.py "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" > $O/pmc_stall.txt 2>> $O/err
rocprofv3 --kernel-include-regex "verify_ddc|window_kernel" --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/p5 -o p -- $PM > /dev/null 2>> $O/err
python $R/scripts/pmc_table.py "$(find /tmp/p5 -name '*counter_collection.csv' | head -1)" > $O/pmc_sca.txt 2>> $O/err
cat $O/pmc_sq.txt $O/pmc_stall.txt $O/pmc_sca.txt; tail -3 $O/err
