#!/bin/bash
# CPU builds of the native pieces under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on the
# pool): the code-generation table (csrc/instructions_table.cpp), the checker (oracle/tc_oracle*.c) and the mex gateway
# (integration/tc_mex.c against tests/mock_mex).  Runs the CPU tests that exercise them.
#   bash scripts/sanitize_cpu.sh > profiles/r05_sanitize_cpu.log 2>&1
# Every finding is FATAL (UBSan halts, ASan aborts) and ALSO written to log files that are printed at the end, so that
# neither pytest's output capture nor a pipe can hide one; the script's exit status is pytest's.
set -u -o pipefail
cd "$(dirname "$0")/.."
LOGDIR=$(mktemp -d)
export TC_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:log_path=$LOGDIR/asan
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$LOGDIR/ubsan
python -c "import __graft_entry__ as g; print(g.build_native_table()); from oracle import build as ob; print(ob.build_oracle())"
python -m pytest -q -x -s -p no:cacheprovider tests/test_instructions_table.py tests/test_csparse_ops.py tests/test_oracle_sls.py \
  tests/test_equilibrium.py tests/test_tape_roundtrip.py tests/test_reference_flops_table.py -m "not gpu"
rc=$?
echo "---- sanitizer log files ($LOGDIR) ----"
n=0
for f in "$LOGDIR"/*; do
  [ -e "$f" ] || continue
  n=$((n + 1))
  echo "== $f"; cat "$f"
done
echo "sanitizer log files: $n (0 = no finding); pytest exit status $rc"
rm -rf "$LOGDIR"
exit $rc
