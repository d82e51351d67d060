# two-stream timeline of one step of the TIMED region (bench.py order: warm-up, instrumented pass, 3-step serial probe, timed region, serial pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -- python $R/bench.py --processes 1 --steps 6 --warmup 2 --cpu-sample 0 > /tmp/prof_tl.log 2>&1
# the serial pass at the end runs max(5, min(steps, 20)) = 6 steps x 2 nl_setup launches; the timed region's last-but-one step starts 4 anchors before that
python $R/tools/rocpd_timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) nl_setup 130 - 16
