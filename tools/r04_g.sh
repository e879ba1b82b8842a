set -u
cd $GRAFT_REPO_ROOT
T=r04_g
MURCL_SEQ_N=400 bash tools/trace_seq.sh ${T}_stage2 $GRAFT_REPO_ROOT/tools/bench_full.py --stage 2 --steps 4 > gpurun_out/${T}_stage2_seq.txt 2>&1
tail -3 gpurun_out/${T}_stage2_seq.txt
