#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_round.sh r01 w8
# Collects, for bench.py's exact command line: (1) rocprofv3 --kernel-trace --stats summary,
# (2) separate PMC passes for FETCH_SIZE, WRITE_SIZE and the SQ stall counters (never combined
# with a trace domain).  Raw output under gpurun_out/, digest via scripts/digest_profile.py.
set -u
ROUND=${1:-r01}; VAR=${2:-w8}
R=$(pwd); OUT=$R/gpurun_out/prof_${ROUND}_${VAR}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-configs --sustain 0 --variant $VAR > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --sustain 0 --variant $VAR > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --sustain 0 --variant $VAR > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --sustain 0 --variant $VAR > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --sustain 0 --variant $VAR > $OUT/pmc_sq2.log 2>&1
cd $R && python3 scripts/digest_profile.py $OUT $ROUND $VAR
