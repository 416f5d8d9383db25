# memory-side PMC passes for the level-0 ICP kernel (level_probe.py, one level)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="scripts/level_probe.py --level 0 --iters 4"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmc_tcp -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_tcp.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/pmc_tcc -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_tcc.err
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d gpurun_out/pmc_ta -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_ta.err
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_lat -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_lat.err
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCC_REQ_sum --output-format csv -d gpurun_out/pmc_ea -- python3 $ARGS > /dev/null 2> gpurun_out/pmc_ea.err
tail -2 gpurun_out/pmc_*.err
