set -u
cd $GRAFT_REPO_ROOT
T=r04_d
MURCL_SEQ_N=60 bash tools/trace_seq.sh ${T}_dsmil $GRAFT_REPO_ROOT/tools/dsmil_seq.py > gpurun_out/${T}_dsmil_seq.txt 2>&1
MURCL_SEQ_N=40 bash tools/trace_seq.sh ${T}_clam $GRAFT_REPO_ROOT/tools/clam_seq.py train > gpurun_out/${T}_clam_train_seq.txt 2>&1
