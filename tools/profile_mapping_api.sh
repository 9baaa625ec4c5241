cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_map
rocprofv3 --kernel-trace --hip-trace --stats -d gpurun_out/prof_map -o map -- python3 tools/bench_mapping.py > gpurun_out/prof_map.log 2>&1
python3 - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_map/map_results.db'); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'top' in t or 'api' in t.lower() or 'region' in t.lower()][:20])
for t in tabs:
    if t.startswith('top'):
        try:
            rows=list(cur.execute(f"select * from {t} limit 14"))
            print(t, [d[0] for d in cur.description])
            for r in rows: print("  ", r[:6])
        except Exception as e: print(t, e)
PY
