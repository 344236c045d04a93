#!/bin/bash
# Same-box A/B of the plan-level Winograd modes (process default AWR_WINOGRAD = 0 direct / 1 forward / 2 forward + data and weight gradients) on the
# bench headline and on Hourglass-1: tools/wino_ab.sh [modes...]   (default: 0 1 2)
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 --no-accurate-mode --no-data-path --no-winograd --steps 20 --warmup 5"
MODES=${@:-0 1 2}
for w in $MODES; do
  AWR_WINOGRAD=$w python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 wino=$w', d['value'], d['ms_per_step'])"
  AWR_WINOGRAD=$w python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 wino=$w', d['value'], d['ms_per_step'])"
done
