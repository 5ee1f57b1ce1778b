set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p2a; mkdir -p $out
python3 tools/run_2a_once.py 1048576 1 10 > $out/plain.txt 2>&1
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pass$i -- python3 tools/run_2a_once.py 1048576 1 3 > $out/pass$i.txt 2> $out/pass$i.err || echo "pass $i failed"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/run_2a_once.py 1048576 1 10 > $out/stats.txt 2>&1
python3 tools/pmc_summary.py $out stg_ > $out/summary.txt; cat $out/plain.txt $out/summary.txt; cat $out/stats/*/*kernel_stats.csv | cut -c1-200
