import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
tr=[(e-s)/1e3 for s,e,n in rows if 'trail_potf2_kernel' in n]
pn=[(e-s)/1e3 for s,e,n in rows if 'panel_oneshot' in n]
# last potrf call = last 31 trail launches
print('trail_potf2 durations of the last factorisation (us):', ' '.join('%.0f'%x for x in tr[-31:]))
print('panel durations (us):', ' '.join('%.1f'%x for x in pn[-32:]))
