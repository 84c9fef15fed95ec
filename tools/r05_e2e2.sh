#!/bin/bash
export TMPDIR=/tmp
for g in dma kernel; do for sp in "" "4096" "1024,3072" "2048,2048"; do echo -n "gather $g split '$sp': "; FRP_HOST_GATHER=$g FRP_HOST_SPLIT=$sp python tools/dbg/e2e_reg.py 2>/dev/null; done; done
