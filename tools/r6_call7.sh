# round 6, GPU call 7: split-role issue probe (MFMA wave | VALU partner on one SIMD); phased 64-row attention (opposite phases) vs product
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6g; mkdir -p $O
timeout 120 tools/lab/issue_probe.bin > $O/issue_probe.txt 2>&1
cat $O/issue_probe.txt
VIDITQ_LIB=$R/vidit-q_amd/csrc/libviditq_hip.so timeout 300 python tools/attn_ab.py --dump=/tmp/attn_ref.pt > $O/attn64p_cmp.txt 2>&1
for d in _ab_attn64p _ab_attn64p4 _ab_attn64p_noprio; do
  VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 120 python tools/attn_ab.py --cmp=/tmp/attn_ref.pt >> $O/attn64p_cmp.txt 2>&1 || echo "$d: rc $?" >> $O/attn64p_cmp.txt
done
cat $O/attn64p_cmp.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_attn64p _ab_attn64p4 _ab_attn64p_noprio; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 120 python tools/attn_ab.py spatial image >> $O/attn64p_ab.txt 2>> $O/attn64p_ab.err
  done
done
cat $O/attn64p_ab.txt
