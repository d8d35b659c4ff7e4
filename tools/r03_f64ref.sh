#!/bin/bash
# float64 rows in the stationary phase: does a rolling refresher (PTG_REFRESH_ALWAYS=1) help once the batch has spread over the tables?
for rep in 1 2; do
  echo "== default"; TS_DTYPE=float64 timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | grep -o "T100.*"
  echo "== PTG_REFRESH_ALWAYS=1"; PTG_REFRESH_ALWAYS=1 TS_DTYPE=float64 timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | grep -o "T100.*"
done
echo "== float32 default"; timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | grep -o "T100.*"
echo "== float32 PTG_REFRESH_ALWAYS=1"; PTG_REFRESH_ALWAYS=1 timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | grep -o "T100.*"
true
