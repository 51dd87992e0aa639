#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_sq.sh r05 dcmotor ldscs
# SQ counter passes (never combined with a trace domain) of  python3 bench.py --config <config> --variant <variant>:
# where the wavefronts' cycles go (issue, waiting, instruction fetch), instruction mix, LDS bank conflicts.
set -u
ROUND=${1:-r05}; CFG=${2:-dcmotor}; VAR=${3:-ldscs}; BATCH=${4:-0}
R=$(pwd); OUT=$R/gpurun_out/sq_${ROUND}_${CFG}_${VAR}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps 2 --warmup 1 --no-cpu --no-calib > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps 2 --warmup 1 --no-cpu --no-calib > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq3 -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps 2 --warmup 1 --no-cpu --no-calib > $OUT/pmc_sq3.log 2>&1
cd $R && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Kernel_Name", "").startswith("tc_solve_kernel"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{d} {k} {sum(v) / len(v):.6g}  (launches {len(v)})")
PY
