# timeline of one step of config 4 (neighbour list 9 A + PME, fp64): every launch with start / end / duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace -d /tmp/prof_c4 -- python $R/bench.py --workload c4 --steps 6 --warmup 2 --cpu-sample 0 > /tmp/prof_c4.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_c4 -name "*.db" | head -1) nl_setup 60 - 3
