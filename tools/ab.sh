#!/bin/bash
# Same-box A/B of the in-tree library against another build (AWR_LIB_PATH): tools/ab.sh <other.so> <reps> <bench args...>
OTHER=$1; REPS=$2; shift 2
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras"
for i in $(seq $REPS); do
  python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  ', d['value'], d['ms_per_step'])"
  AWR_LIB_PATH=$OTHER python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('other', d['value'], d['ms_per_step'])"
done
