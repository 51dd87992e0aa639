#!/bin/bash
# CPU builds of the native pieces under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on the
# pool): the code-generation table (csrc/instructions_table.cpp), the checker (oracle/tc_oracle*.c) and the mex gateway
# (integration/tc_mex.c against tests/mock_mex).  Runs the CPU tests that exercise them; prints the sanitizer reports.
#   bash scripts/sanitize_cpu.sh > profiles/r04_sanitize_cpu.log 2>&1
set -u
cd "$(dirname "$0")/.."
export TC_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1
python -c "import __graft_entry__ as g; print(g.build_native_table()); from oracle import build as ob; print(ob.build_oracle())"
python -m pytest -q -x -p no:cacheprovider tests/test_instructions_table.py tests/test_csparse_ops.py tests/test_oracle_sls.py \
  tests/test_equilibrium.py tests/test_tape_roundtrip.py -m "not gpu" 2>&1 | tail -15
echo "sanitizer reports above (none = clean); exit $?"
