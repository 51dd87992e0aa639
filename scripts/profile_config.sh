#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_config.sh r02 quadcopter ldsm4 [steps] [batch]
# rocprofv3 --kernel-trace --stats summary + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of
#   python3 bench.py --config <config> --variant <variant> --steps 3 --warmup 1 --no-cpu
set -u
ROUND=${1:-r02}; CFG=${2:-quadcopter}; VAR=${3:-ldsm4}; STEPS=${4:-3}; BATCH=${5:-0}
R=$(pwd); OUT=$R/gpurun_out/prof_${ROUND}_${CFG}_${VAR}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps $STEPS --warmup 1 --no-cpu --no-calib > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps 2 --warmup 1 --no-cpu --no-calib > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --variant $VAR --batch $BATCH --steps 2 --warmup 1 --no-cpu --no-calib > $OUT/pmc_write.log 2>&1
cd $R && TC_PROFILE_BATCH=$BATCH TC_PROFILE_STEPS=$STEPS python3 scripts/digest_profile.py $OUT $ROUND ${CFG}_${VAR}
