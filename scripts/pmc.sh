# PMC passes for the ICP bench (counters only: no trace domains alongside --pmc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-extras --cpu-pairs 0 --pairs-per-gpu ${PAIRS:-64}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_misc -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_misc.err
ls gpurun_out/pmc_*/*/ | head -30
