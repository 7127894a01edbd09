cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 4 --warmup 2 --workload config2 --no-cpu-baseline --no-next-rows"
for v in ${VARIANTS:-2 1}; do
  if [ $v = 2 ]; then export GSR_BWD2=1; fi
  O=gpurun_out/pmcq$v; mkdir -p $O
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o p1 -- $CMD > $O/p1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o p2 -- $CMD > $O/p2.log 2>&1
  python tools/pmc_summary.py $O config2 2>/dev/null | grep -E "kernel|blend_" | tee -a gpurun_out/pmcq.txt
  find $O -name "*.db" -delete
done
