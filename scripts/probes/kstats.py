import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv)>2 else 12]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
