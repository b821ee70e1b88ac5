import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
tot=sum(int(r[2]) for r in rows[1:])
print("total ns",tot)
for r in rows[1:45]:
    print(f"{r[0][:100]:100s} calls {r[1]:>5s} avg {float(r[3])/1e3:8.1f} us  {r[4]:>6s}%")
