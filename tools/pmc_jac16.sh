#!/bin/bash
# PMC passes over the float16 band Jacobian at 64 crops per launch (tools/jac16_time.py 64): tools/pmc_jac16.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}
bash $R/tools/pmc_any.sh jac16a_$TAG tools/jac16_time.py SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
bash $R/tools/pmc_any.sh jac16b_$TAG tools/jac16_time.py SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
bash $R/tools/pmc_any.sh jac16c_$TAG tools/jac16_time.py TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
bash $R/tools/pmc_any.sh jac16d_$TAG tools/jac16_time.py SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS
