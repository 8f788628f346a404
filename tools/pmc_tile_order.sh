set -e
ROOT=$(pwd); export TMPDIR=/tmp
for ws in 0 1 2; do
  out=$ROOT/gpurun_out/pmc_tile_$ws; mkdir -p $out
  (cd /tmp && TILE_WS=$ws rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o t -- python $ROOT/tools/tile_order_bench.py > $out/log.txt 2>&1) || { tail -3 $out/log.txt; }
  DB=$(ls $out/*.db 2>/dev/null | head -1)
  python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
rows = db.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name").fetchall()
for n, c, v in rows:
    if "gemm16" in n: print("wstat $ws", n[:60], "launches", c, "fetched MB per launch (x2 corrected)", round(2 * v / c * 1024 / 1e6, 1))
PY
  rm -rf $out
done
