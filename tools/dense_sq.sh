# SQ counters of the large-context path's kernels (cfg #5, fp64):  bash tools/dense_sq.sh <tag> [kernel substrings...]
tag=${1:-rXX}; shift; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/dsq/p1 -- python3 tools/dense_profile.py f64 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $out/dsq/p2 -- python3 tools/dense_profile.py f64 3 > /dev/null 2>&1
for k in "${@:-ztz trtri_ll chol_ll dense_grad_tile}"; do for kk in $k; do echo "== $kk"; python tools/pmc_kernel.py $out/dsq $kk; done; done > $out/dense_sq_counters.txt 2>&1
rm -rf $out/dsq; cat $out/dense_sq_counters.txt
