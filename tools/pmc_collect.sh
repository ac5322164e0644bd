#!/bin/bash
# rocprofv3 passes over tools/kall.py (one representative launch per kernel family), as MI355X_MICROARCH.md prescribes: kernel trace +
# stats in one run, every --pmc group in its OWN run (never combined with sys / hip / memory tracing).  Usage (on the GPU box, from
# the repo root):  bash tools/pmc_collect.sh gpurun_out/pmc_r03     then  python tools/pmc_summarize.py gpurun_out/pmc_r03 profiles/r03_pmc_summary.json
set -u
OUT=$(realpath -m "${1:-gpurun_out/pmc_r03}")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export KALL_INFO="$OUT/cases.json"
run() { tag=$1; shift; timeout 400 rocprofv3 --kernel-trace "$@" -d "$OUT" -o "$tag" --output-format csv -- python "$REPO/tools/kall.py" > "$OUT/$tag.log" 2>&1 || echo "pass $tag failed" >> "$OUT/errors.log"; }
run stats --stats
run f --pmc FETCH_SIZE
run w --pmc WRITE_SIZE
run h --pmc TCC_HIT_sum TCC_MISS_sum
run m --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run s --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run l --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
ls -la "$OUT" > "$OUT/ls.txt"
