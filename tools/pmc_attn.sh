#!/bin/bash
# Issue-level counters of the attention kernels (tools/attnone.py shapes), one --pmc group per rocprofv3 run.
# Usage on the GPU box, from the repo root: bash tools/pmc_attn.sh gpurun_out/pmc_attn [attnone args]
set -u
OUT=$(realpath -m "${1:-gpurun_out/pmc_attn}"); shift || true
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
run() { tag=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT" -o "$tag" --output-format csv -- python "$REPO/tools/attnone.py" --views 96 --reps 2 ${ATTNONE_ARGS:-} > "$OUT/$tag.log" 2>&1 || echo "pass $tag failed" >> "$OUT/errors.log"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
run c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
run d SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE
run e SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:36s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
