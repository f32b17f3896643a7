# A/B of library builds + stage attribution (GPU box): bash tools/r3_ab.sh <tag> lib1 lib2 ...
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
L=real-time-neural-rendering-of-lidar-point-clouds_amd/lib
LIBS=""
for v in "$@"; do LIBS="$LIBS $L/librtr_hip${v:+_}$v.so"; done
timeout -k 10 400 python tools/ab_frame.py $LIBS > $O/ab.txt 2>&1; rc=$?; cat $O/ab.txt; [ $rc -ge 124 ] && exit 1
if [ -n "$AB_XP" ]; then
RTR_LIB_VARIANT=xp timeout -k 10 240 python tools/kbench.py --scenes room_shell --frames 24 \
  --options "xp=0;xp=128;xp=256;xp=64;xp=8;xp=4;xp=0" > $O/xp.jsonl 2> $O/xp.err
python3 -c "
import sys,json
for l in open('$O/xp.jsonl'):
    d=json.loads(l); print(d['opts'].ljust(16), 'T1', d['min_depth'], 'tile', d['tile'], 'filter', d['filter'], 'probe', d['probe'])
"
fi
