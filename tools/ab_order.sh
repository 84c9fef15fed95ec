#!/bin/bash
# A/B of the queue-order key weights (FRP_ORDER_W=w_eq,w_in; 0,0 = objective of the initial guess alone, the round-2 key)
for w in "0,0" "40,20" "40,0" "0,20" "20,20" "80,40" "200,100"; do
  for c in 2 3; do
    FRP_ORDER_W=$w python bench.py --config $c --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w=$w config $c', round(j['value']), round(j['ms_per_step'],4))"
  done
done
