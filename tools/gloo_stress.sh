#!/bin/bash
# tests/test_dist_gloo.py N times (default 20) under CPU load (one busy loop per core) -- the condition under which the round-5 form of
# the test lost the fd-sharing race.  Prints one line per run and a tally; exit 0 only if every run passed.
n=${1:-20}
cores=$(nproc)
pids=()
for i in $(seq "$cores"); do (while :; do :; done) & pids+=($!); done
trap 'kill "${pids[@]}" 2>/dev/null' EXIT
ok=0
for i in $(seq "$n"); do
  if out=$(python -m pytest tests/test_dist_gloo.py -x -q -p no:cacheprovider 2>&1 | tail -1); then :; fi
  echo "run $i: $out"
  case "$out" in *passed*) case "$out" in *failed*) ;; *) ok=$((ok+1));; esac;; esac
done
echo "gloo stress: $ok / $n green under $cores busy loops"
[ "$ok" = "$n" ]
