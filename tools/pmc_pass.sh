cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_$tag -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --overlap 0 > /tmp/prof_$tag.log 2>&1
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $db $R/gpurun_out/pmc2/$tag.csv 2>&1 | tail -2
done
ls -la $R/gpurun_out/pmc2
