#!/bin/bash
export TMPDIR=/tmp
python tools/twist_latency.py 2>/dev/null | grep "B     1\|B   256\|B  1024\|dropin"
bash tools/r05_gputests.sh
