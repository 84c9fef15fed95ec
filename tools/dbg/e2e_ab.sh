#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "registered_buffers or host_path" 2>&1 | tail -5
timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(j['end_to_end'], indent=1)[:1500]); print('value', j['value'], j['ms_per_step'])"
