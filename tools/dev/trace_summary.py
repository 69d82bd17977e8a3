"""Per-kernel (name, grid) durations from a rocprofv3 rocpd database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tr/tr_results.db')
rows = db.execute('select name, count(*), sum(end-start), avg(end-start), grid_x, workgroup_x '
                  'from kernels group by name, grid_x order by sum(end-start) desc limit %d'
                  % (int(sys.argv[2]) if len(sys.argv) > 2 else 30)).fetchall()
for r in rows:
    print('%-72s calls %4d total_us %8d avg_us %8.1f grid %d wg %d' % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4], r[5]))
