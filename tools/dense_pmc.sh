# HBM traffic of the large-context path's kernels (cfg #5, fp64): separate FETCH_SIZE / WRITE_SIZE passes over tools/dense_profile.py
#   bash tools/dense_pmc.sh r04      ->  gpurun_out/r04/dense_pmc_hbm_traffic.{json,txt}
tag=${1:-rXX}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/dpmc/fetch -- python3 tools/dense_profile.py f64 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/dpmc/write -- python3 tools/dense_profile.py f64 4 > /dev/null 2>&1
python tools/pmc_summary.py $out/dpmc $out/dense_pmc_hbm_traffic.json 4 > $out/dense_pmc_hbm_traffic.txt
rm -rf $out/dpmc; cat $out/dense_pmc_hbm_traffic.txt
