# T1 / tile / frame device times of several library builds on one box, two alternating rounds:
#   tools/ab_t1.sh <variant|default> ...      (librtr_hip_<variant>.so, RTR_LIB_VARIANT; AB_OPTS = kbench option sets)
R=${GRAFT_REPO_ROOT:-.}
OPTS=${AB_OPTS:-"pack=1;pack=1"}
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset RTR_LIB_VARIANT; else export RTR_LIB_VARIANT=$v; fi
  timeout -k 10 180 python $R/tools/kbench.py --scenes ${AB_SCENE:-room_shell} --points ${AB_N:-100000000} --frames 40 --filter ${AB_FILTER:-1} --options "$OPTS" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if not line.startswith('{'): continue
    d = json.loads(line); print('$v'.ljust(10), d['opts'].ljust(24), 'T1', d.get('min_depth'), 'tile', d.get('tile'), 'filter', d.get('filter'), 'frame', d['frame_us'])"
done
done
