import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in con.execute("pragma table_info(kernels)")]
print(cols)
want = [c for c in ("name", "grid_x", "workgroup_x", "lds_size", "vgpr_count", "accum_vgpr_count", "scratch_size", "duration") if c in cols]
seen = set()
for row in con.execute(f"select {', '.join(want)} from kernels where name like '%dgpu%'"):
    key = row[:-1]
    if key in seen: continue
    seen.add(key)
    print(dict(zip(want, [str(x)[:60] for x in row])))
