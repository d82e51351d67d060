cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -- python $R/bench.py --processes 1 --steps 6 --warmup 2 --cpu-sample 0 > /tmp/prof_tl.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) nl_setup 110
