#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/ keeps for round 4 (kernel-trace stats + PMC passes of the headline command, C5 evidence for
# both precisions, every bench line).  Then here: python tools/collect_profiles.py r04; python tools/collect_c5.py r04 r4c5_bf16 bf16;
# python tools/collect_c5.py r04 r4c5_f32 f32
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh final > gpurun_out/final_prof.txt 2>&1
bash tools/profile_pmc.sh final_fetch "FETCH_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_write "WRITE_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh final_sq "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" > /dev/null 2>&1
bash tools/profile_pmc.sh final_inst "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" > /dev/null 2>&1
bash tools/profile_pmc.sh final_valu "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" > /dev/null 2>&1
bash tools/profile_pmc.sh final_clk "GRBM_GUI_ACTIVE" > /dev/null 2>&1
bash tools/r4_c5_prof.sh bf16 > gpurun_out/final_c5_bf16.txt 2>&1
bash tools/r4_c5_prof.sh f32 > gpurun_out/final_c5_f32.txt 2>&1
bash tools/run_final_benches.sh > gpurun_out/final_benches.txt 2>&1
tail -45 gpurun_out/final_benches.txt
