import csv, sys, glob
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'].split('(')[0].replace('void ndlqr::',''), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'], r['VGPR_Count'], r['LDS_Block_Size']) for r in rows]
ks=[x for x in seq if 'amd_rocclr' not in x[0]]
n=int(sys.argv[2])
for x in ks[-n:]: print('%-44s %8.1f us grid=%-7s vgpr=%s lds=%s'%x)
