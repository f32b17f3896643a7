# T1 of several library builds, one after the other on one box: tools/ab_variants.sh <variant> ...
# (librtr_hip_<variant>.so, see RTR_LIB_VARIANT in _lib.py); two rounds so that box drift shows
R=$GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  RTR_LIB_VARIANT=$v timeout -k 10 120 python $R/tools/kbench.py --scenes room_shell --frames 30 --options "pack=1;pack=1" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v'.ljust(10), 'T1', d.get('min_depth'), 'tile', d.get('tile'), 'frame', d['frame_us'])"
done
done
