set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo "default       " $(python tools/bench_rows.py 2>/dev/null | tail -1)
echo "GROUP_WGRAD=0 " $(MURCL_GROUP_WGRAD=0 python tools/bench_rows.py 2>/dev/null | tail -1)
echo "CLAM_POOL2=0  " $(MURCL_CLAM_POOL2=0 python tools/bench_rows.py 2>/dev/null | tail -1)
echo "both off      " $(MURCL_GROUP_WGRAD=0 MURCL_CLAM_POOL2=0 python tools/bench_rows.py 2>/dev/null | tail -1)
done
MURCL_SEQ_N=44 bash tools/trace_seq.sh r04_l_clam_inst $GRAFT_REPO_ROOT/tools/clam_seq.py inst > gpurun_out/r04_l_clam_inst_seq.txt 2>&1
tail -46 gpurun_out/r04_l_clam_inst_seq.txt | cut -c1-150
