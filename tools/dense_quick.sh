# cfg #5 in one call: phase stamps of the Cholesky (variant library built with -DPACOH_CHOL_STAMPS=1), pass time, parity tests, per-kernel times
out=gpurun_out/$1; mkdir -p $out
[ -f meta_learning_pacoh_amd/lib/libpacoh_gp_cst.so ] && PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_cst.so python tools/dense_profile.py f64 3 2>&1 | tail -2 > $out/st.txt; [ -f $out/st.txt ] && cat $out/st.txt
python bench.py --config 5 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 ms per pass', d['ms_per_step'])"
python -m pytest tests/test_gpu_dense_path.py -x -q -m gpu > $out/t.txt 2>&1; tail -1 $out/t.txt
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 tools/dense_profile.py > /dev/null 2>&1
cp $(ls $out/p/*/*kernel_stats.csv | head -1) $out/dense_stats.csv; rm -rf $out/p
python - <<PY
import csv
for r in list(csv.DictReader(open('$out/dense_stats.csv')))[:7]: print('%-70s %4s %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
