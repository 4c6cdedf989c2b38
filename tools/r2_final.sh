#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/ keeps for this round (kernel-trace stats, four PMC passes, bench lines).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh final > gpurun_out/final_prof.txt 2>&1
bash tools/profile_pmc.sh final_fetch "FETCH_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_write "WRITE_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_sq "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" > /dev/null 2>&1
bash tools/profile_pmc.sh final_inst "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" > /dev/null 2>&1
bash tools/profile_pmc.sh final_clk "GRBM_GUI_ACTIVE" > /dev/null 2>&1
bash tools/run_final_benches.sh > gpurun_out/final_benches.txt 2>&1
tail -30 gpurun_out/final_benches.txt
