#!/usr/bin/env bash
# CPU sanitizers on the host-side native code (SURVEY.md section 5 "race detection / sanitizers: run once").
# GPU sanitizers are not available on this pool; nothing here touches a GPU.
#
#   1. ASan + UBSan   rc_tree.cpp (C++ / OpenMP: the lockstep search's host trees, mirroring /root/reference/mcts.py:52-154)
#                     and oracle/rc_oracle.c (C / OpenMP: the CPU checker), built with gcc -fsanitize=address,undefined:
#                       a) tests/abi/tree_driver.cpp linked against the instrumented rc_tree.cpp (no Python in the process)
#                       b) tests/test_tree_native.py and tests/test_oracle.py with the instrumented .so files loaded into Python
#                          (LD_PRELOAD of libasan; leak checking off: CPython itself never frees everything)
#   2. TSan           rc_tree.cpp + tree_driver.cpp built with clang++ -fsanitize=thread -fopenmp (LLVM's libomp, whose OMPT tool
#                     library libarcher tells TSan about OpenMP's barriers and task edges; gcc's libgomp is not annotated and floods
#                     TSan with false positives), OMP_NUM_THREADS=4: per-root generators and the shared-generator mode.
#
# Output: everything under $OUT (default /tmp/rc_sanitize); a summary on stdout, which profiles/r06_sanitizers.txt keeps.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${OUT:-/tmp/rc_sanitize}"
LLVM="${LLVM:-/opt/rocm/lib/llvm}"
mkdir -p "$OUT"
cd "$ROOT"
TREE_ID=$(python3 -c "import importlib.util,sys; s=importlib.util.spec_from_file_location('b','rubiks-cube-solver_amd/_build.py'); m=importlib.util.module_from_spec(s); s.loader.exec_module(m); print(m.source_hash(m.TREE_SOURCES))")
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
fail=0
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
say "== CPU sanitizers, $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1), $($LLVM/bin/clang++ --version | head -1)"

say "-- 1a. ASan + UBSan: tree_driver + rc_tree.cpp (g++ $SAN -fopenmp), 4 threads"
g++ $SAN -std=c++17 -fopenmp -ffp-contract=off -Wall -Wextra -DRC_SRC_HASH=$TREE_ID -Iinclude tests/abi/tree_driver.cpp rubiks-cube-solver_amd/csrc/rc_tree.cpp -o "$OUT/tree_driver_asan" || fail=1
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 "$OUT/tree_driver_asan" 160 60 4 > "$OUT/1a.log" 2>&1
rc=$?; say "   exit $rc: $(tail -1 "$OUT/1a.log")"; [ $rc -eq 0 ] || fail=1
say "   reports: $(grep -c -E 'ERROR: AddressSanitizer|runtime error:|LeakSanitizer' "$OUT/1a.log")"

say "-- 1b. ASan + UBSan: librubiktree.so and librc_oracle.so instrumented, loaded by the CPU tests"
g++ $SAN -std=c++17 -fopenmp -ffp-contract=off -Wall -Wextra -fPIC -shared -DRC_SRC_HASH=$TREE_ID -o "$OUT/librubiktree_asan.so" rubiks-cube-solver_amd/csrc/rc_tree.cpp || fail=1
gcc $SAN -std=c11 -fopenmp -Wall -Wextra -fPIC -shared -o "$OUT/librc_oracle_asan.so" oracle/rc_oracle.c || fail=1
LIBASAN=$(gcc -print-file-name=libasan.so); LIBUBSAN=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$LIBASAN:$LIBUBSAN" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 \
  RUBIKTREE_LIB="$OUT/librubiktree_asan.so" RC_ORACLE_LIB="$OUT/librc_oracle_asan.so" \
  python3 -m pytest tests/test_tree_native.py tests/test_oracle.py -x -q -p no:cacheprovider > "$OUT/1b.log" 2>&1
rc=$?; say "   exit $rc: $(grep -E 'passed|failed|error' "$OUT/1b.log" | tail -1)"; [ $rc -eq 0 ] || fail=1
say "   reports: $(grep -c -E 'ERROR: AddressSanitizer|runtime error:' "$OUT/1b.log")"

say "-- 2. TSan: tree_driver + rc_tree.cpp (clang++ -fsanitize=thread -fopenmp, libomp + libarcher), OMP_NUM_THREADS=4"
"$LLVM/bin/clang++" -fsanitize=thread -fno-omit-frame-pointer -g -O1 -std=c++17 -fopenmp -ffp-contract=off -Wall -Wextra -DRC_SRC_HASH=$TREE_ID -Iinclude \
  tests/abi/tree_driver.cpp rubiks-cube-solver_amd/csrc/rc_tree.cpp -o "$OUT/tree_driver_tsan" -Wl,-rpath,"$LLVM/lib" > "$OUT/2.build.log" 2>&1 || { fail=1; say "   build failed: $(tail -3 "$OUT/2.build.log")"; }
if [ -x "$OUT/tree_driver_tsan" ]; then
  OMP_TOOL_LIBRARIES="$LLVM/lib/libarcher.so" ARCHER_OPTIONS="verbose=1" TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" OMP_NUM_THREADS=4 \
    "$OUT/tree_driver_tsan" 160 60 4 > "$OUT/2.log" 2>&1
  rc=$?; say "   exit $rc (66 = TSan printed a report): $(grep 'tree_driver' "$OUT/2.log" | tail -1)"
  say "   archer: $(grep -i -m1 'archer' "$OUT/2.log")"
  raw=$(grep -c 'WARNING: ThreadSanitizer' "$OUT/2.log")
  ours=$(awk '/WARNING: ThreadSanitizer/{blk=""} {blk=blk"\n"$0} /^SUMMARY: ThreadSanitizer/{if (blk ~ /rc_tree\.cpp/) n++} END{print n+0}' "$OUT/2.log")
  say "   reports: $raw raw, $ours with a frame in rc_tree.cpp (the rest lie inside libomp.so: tools/tsan_libomp.supp)"
  [ "$ours" -eq 0 ] || fail=1
  OMP_TOOL_LIBRARIES="$LLVM/lib/libarcher.so" TSAN_OPTIONS="halt_on_error=0 suppressions=$ROOT/tools/tsan_libomp.supp" OMP_NUM_THREADS=4 \
    "$OUT/tree_driver_tsan" 160 60 4 > "$OUT/2s.log" 2>&1
  rc=$?; say "   with the libomp suppressions: exit $rc, reports: $(grep -c 'WARNING: ThreadSanitizer' "$OUT/2s.log")"; [ $rc -eq 0 ] || fail=1
  # positive control: the same toolchain must SEE a race that is there (an unsynchronised counter in a parallel for)
  printf '#include <cstdio>\nint main(){long c=0;\n#pragma omp parallel for num_threads(4)\nfor(int i=0;i<100000;++i)c+=i;\nstd::printf("%%ld\\n",c);}\n' > "$OUT/control.cpp"
  "$LLVM/bin/clang++" -fsanitize=thread -g -O1 -fopenmp "$OUT/control.cpp" -o "$OUT/control_tsan" -Wl,-rpath,"$LLVM/lib" > /dev/null 2>&1
  OMP_TOOL_LIBRARIES="$LLVM/lib/libarcher.so" TSAN_OPTIONS="halt_on_error=0 suppressions=$ROOT/tools/tsan_libomp.supp" "$OUT/control_tsan" > "$OUT/control.log" 2>&1
  ctl=$(grep -c 'WARNING: ThreadSanitizer: data race' "$OUT/control.log")
  say "   positive control (racy counter in a parallel for): $ctl report(s) -- the toolchain sees races"; [ "$ctl" -ge 1 ] || fail=1
fi
say "== $( [ $fail -eq 0 ] && echo CLEAN || echo FINDINGS ) (logs in $OUT)"
exit $fail
