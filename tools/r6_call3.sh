# round 6, GPU call 3: full GPU suite on the lean library; temporal attention + quantizer ablations
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6c; mkdir -p $O
for r in 1 2; do
  for d in vidit-q_amd/csrc _ab_tq_abl1 _ab_tq_abl2; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py temporal >> $O/temporal_abl.txt 2>> $O/temporal_abl.err
  done
done
cat $O/temporal_abl.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
