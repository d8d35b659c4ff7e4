#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_graph_replay.py -m gpu -q -x > $O/replay.log 2>&1; echo "replay rc=$?"; tail -15 $O/replay.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
bash tools/r03_ab_quick.sh head product
true
