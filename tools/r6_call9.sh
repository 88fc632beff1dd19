# round 6, GPU call 9: issue probe with priorities / 3 waves per SIMD; per-phase priority arms of the phased attention kernel (stamps + timing)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6i; mkdir -p $O
timeout 120 tools/lab/issue_probe.bin > $O/issue_probe.txt 2>&1
grep -A1 "split\|alone" $O/issue_probe.txt
for d in _ab_attn64p_p0 _ab_attn64p_pv _ab_attn64p_pm; do
  echo "== $d" >> $O/attn_stamps.txt
  VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 120 python tools/attn_stamps.py >> $O/attn_stamps.txt 2>&1
done
cat $O/attn_stamps.txt
for r in 1 2; do
  for d in vidit-q_amd/csrc _ab_attn64p_p0 _ab_attn64p_pv _ab_attn64p_pm; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 120 python tools/attn_ab.py spatial image >> $O/attn64p_ab.txt 2>> $O/attn64p_ab.err
  done
done
cat $O/attn64p_ab.txt
