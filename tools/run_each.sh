#!/bin/bash
# run every test id of the given files in its own pytest process (a crash / segfault then costs one test, and its
# traceback is kept): tools/run_each.sh <logfile> <pytest file / id args...>
log=$1; shift
: > "$log"
ids=$(python -m pytest "$@" -m gpu --collect-only -q 2>/dev/null | grep "::")
for id in $ids; do
  out=$(timeout 900 python -m pytest "$id" -q -x --tb=short -p no:cacheprovider 2>&1)
  rc=$?
  if [ $rc -eq 0 ]; then echo "PASS $id" >> "$log"; else
    echo "FAIL($rc) $id" >> "$log"
    echo "$out" | grep -v "amdgpu.ids\|socket.cpp\|^$" | tail -45 >> "$log"
    echo "----" >> "$log"
  fi
done
grep -c "^PASS" "$log"; grep "^FAIL" "$log"
