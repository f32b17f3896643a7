# one rocprofv3 --pmc pass of a short bench run (GPU box): bash tools/pmc_pass.sh <tag> "<counters>" [bench args]
R=$GRAFT_REPO_ROOT
TAG=$1; CTR=$2; shift; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $CTR --output-format csv -d $O/pmc_x -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-parity "$@" > $O/bench.json 2> $O/err.txt || exit 1
python3 $R/tools/pmc_summary.py $O
