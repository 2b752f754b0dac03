import csv, sys, glob
f=glob.glob(sys.argv[1]+'/**/*_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
print(len(idx), "adam launches")
for k in range(max(0,len(idx)-8), len(idx)-1):
    a,b=idx[k],idx[k+1]
    step=rows[a:b+1]
    t0=int(step[0]['Start_Timestamp'])
    cur_end=int(step[0]['End_Timestamp'])
    prevname=step[0]['Kernel_Name'][:30]
    gaps=[]; idle=0
    for r in step[1:]:
        s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
        if s>cur_end:
            idle+=s-cur_end
            if (s-cur_end)>8000: gaps.append((round((s-cur_end)/1e3,1), round((cur_end-t0)/1e3), prevname[:24], r['Kernel_Name'][:24]))
        if e>cur_end: cur_end=e; prevname=r['Kernel_Name'][:30]
    print("step", k, "len %.1f us idle %.1f us"%((int(step[-1]['Start_Timestamp'])-t0)/1e3, idle/1e3), gaps)
