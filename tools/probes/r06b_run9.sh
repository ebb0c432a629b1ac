cd /root/repo
export TMPDIR=/tmp
bash tools/c4_ab.sh "default:A3T_X=0" "tt_whenever_legal:A3T_GEMM_TT=1" "default:A3T_X=0" "tt_whenever_legal:A3T_GEMM_TT=1" "tt_but_dq_two_launches:A3T_GEMM_TT=1 A3T_ATTN_DQ_DUAL=0"
