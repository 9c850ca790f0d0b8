#!/bin/bash
# PMC pass over the weights-stationary variants on layer3's conv3 (tiles 91 / 94 / 95 / 96), 15 frames
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_ws
mkdir -p $O
CONV="tests/bench_conv.py --batch 15 --shapes l3.c3 --tiles 91,94,95,96 --reps 3"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  -d $O/sq -o p --output-format csv -- python3 $CONV --manifest $O/manifest.json > $O/sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  -d $O/sq2 -o p --output-format csv -- python3 $CONV > $O/sq2.log 2>&1
python3 - <<PY
import csv,collections
for d in ("sq","sq2"):
    rows=list(csv.DictReader(open(f"gpurun_out/pmc_ws/{d}/p_counter_collection.csv")))
    by=collections.OrderedDict()
    for r in rows:
        k=(r["Kernel_Name"][:60], r["Dispatch_Id"])
        by.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
    # last dispatch per kernel name
    last={}
    for (name,did),v in by.items():
        if "conv1x1_ws" in name: last[name]=v
    for name,v in last.items():
        print(d, name, {k:int(x) for k,x in v.items()})
PY
