import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = int(rows[0]["Start_Timestamp"])
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], (int(r["Start_Timestamp"]) - t0) / 1e6) for r in rows]
big.sort(reverse=True)
for d, f, t in big[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]: print(f"{d/1e6:9.3f} ms  {f:32s} at {t:10.1f} ms")
