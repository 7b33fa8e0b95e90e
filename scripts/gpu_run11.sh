#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/t11_pmc1.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc1 > $GRAFT_REPO_ROOT/gpurun_out/pmc1_summary.csv 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/t11_pmc2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc2 > $GRAFT_REPO_ROOT/gpurun_out/pmc2_summary.csv 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc3 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/t11_pmc3.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc3 > $GRAFT_REPO_ROOT/gpurun_out/pmc3_summary.csv 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc4 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/t11_pmc4.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc4 > $GRAFT_REPO_ROOT/gpurun_out/pmc4_summary.csv 2>&1
cd $GRAFT_REPO_ROOT; head -5 gpurun_out/pmc1_summary.csv | cut -c1-300; tail -3 gpurun_out/t11_pmc2.log | cut -c1-200
