#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/ keeps for round 3 (kernel-trace stats, PMC passes, bench lines, C5 evidence).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh final > gpurun_out/final_prof.txt 2>&1
bash tools/profile_pmc.sh final_fetch "FETCH_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_write "WRITE_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_sq "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" > /dev/null 2>&1
bash tools/profile_pmc.sh final_inst "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" > /dev/null 2>&1
bash tools/profile_pmc.sh final_valu "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" > /dev/null 2>&1
bash tools/profile_pmc.sh final_clk "GRBM_GUI_ACTIVE" > /dev/null 2>&1
bash tools/r3_c5c.sh > gpurun_out/final_c5c.txt 2>&1
bash tools/run_final_benches.sh > gpurun_out/final_benches.txt 2>&1
tail -40 gpurun_out/final_benches.txt
