# round 6, GPU call 23: all GEMM waves at s_setprio 2 (above the other stream's co-resident quantizer waves) vs product
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6u; mkdir -p $O
bash tools/ab_env.sh $O/ab 3 "base:" "prio2:VIDITQ_LIB=$R/_ab_gemm_prio/libviditq_hip.so" > $O/ab.txt 2>&1
cat $O/ab.txt
