#!/bin/bash
# Run ON THE GPU BOX: SQ stall counters of one config's solve kernel (separate --pmc passes, no trace domain).
#   bash scripts/profile_sq.sh quadcopter ldsc4
set -u
CFG=${1:-quadcopter}; VAR=${2:-ldsc4}
R=$(pwd); OUT=$R/gpurun_out/sq_${CFG}_${VAR}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH --output-format csv -d $OUT/a -- python3 $R/bench.py --config $CFG --variant $VAR --steps 2 --warmup 1 --no-cpu > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/bench.py --config $CFG --variant $VAR --steps 2 --warmup 1 --no-cpu > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/c -- python3 $R/bench.py --config $CFG --variant $VAR --steps 2 --warmup 1 --no-cpu > $OUT/c.log 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("tc_solve_kernel"): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%s,%.6g" % (k, sum(agg[k])/len(agg[k])))
PY
