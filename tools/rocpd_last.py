"""Print the kernels of the last N dispatches of a rocprofv3 rocpd database (name, duration)."""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
for nme, s, e in rows[-n:]:
    print("%-72s %8.3f ms" % (re.sub(r'\(.*', '', nme)[:72], (e - s) / 1e6))
