set -u
cd $GRAFT_REPO_ROOT
T=r04_h
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "grouped or gate or softmax or pool or clam" 2>&1 | tail -5 > gpurun_out/${T}_tests_kernels.log
python -m pytest tests/test_gpu_modules.py tests/test_gpu_supervised_steps.py tests/test_gpu_fullsize.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${T}_tests_modules.log
cat gpurun_out/${T}_tests_kernels.log gpurun_out/${T}_tests_modules.log
MURCL_SEQ_N=36 bash tools/trace_seq.sh ${T}_clam $GRAFT_REPO_ROOT/tools/clam_seq.py train > gpurun_out/${T}_clam_train_seq.txt 2>&1
tail -30 gpurun_out/${T}_clam_train_seq.txt | cut -c1-140
for L in default pgfuse3 pgexact default pgfuse3 pgexact; do
  if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
  python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_$L.json 2>> gpurun_out/${T}_bench.err
  python -c "
import json
d=json.load(open('gpurun_out/${T}_bench_$L.json')); k=d['kernel_ms_per_step']; print('$L', d['value'], d['ms_per_step'], d['step_stats']['median_ms'], k.get('panel_gemm<K512,BIAS_RELU>'), k.get('panel_gemm<K512,MASK>'), d['rows']['clam_sb_c3_fwd_bwd_aggregator']['ms'], d['rows']['clam_sb_c3_fwd_bwd_aggregator_training_mode']['ms'], d['rows']['clam_sb_c3_fwd_bwd_instance_loss']['ms'], d['rows']['dsmil_c5_share_fwd_bwd']['ms'])
"
done
