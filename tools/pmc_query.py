import sqlite3, collections, sys, glob
for dbn in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    db = sqlite3.connect(dbn); cur = db.cursor()
    try:
        rows = cur.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection").fetchall()
    except Exception as e:
        print(dbn, e); continue
    byd = collections.defaultdict(dict)
    for did, kn, gs, cn, v, dur in rows:
        byd[did]['name'] = kn[:50]; byd[did]['grid'] = gs; byd[did]['dur_ns'] = dur
        byd[did][cn] = byd[did].get(cn, 0) + v
    pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
    for did in sorted(byd):
        d = byd[did]
        if pat in d['name']:
            print(dbn.split('/')[-1], {k: (round(v) if isinstance(v, float) else v) for k, v in d.items()})
