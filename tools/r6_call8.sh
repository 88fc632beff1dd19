# round 6, GPU call 8: split-role issue probe with role-specific loops; phase stamps of the phased attention kernel
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
timeout 120 tools/lab/issue_probe.bin > $O/issue_probe.txt 2>&1
grep -A1 split $O/issue_probe.txt
VIDITQ_LIB=$R/_ab_attn64p_st/libviditq_hip.so timeout 120 python tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
cat $O/attn_stamps.txt
